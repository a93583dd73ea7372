mkdir -p gpurun_out/r03w
for r in 56 40 24 12; do
  echo "== NEUTRAL_STREAM_REFILL=$r"
  python tools/ablate.py matrix --libs default --env NEUTRAL_STREAM_REFILL=$r --run "csp 400 100000000 10 2" --run "stream 400 10000000 1 2" --run "stream 4000 1000000 1 2" 2>&1 | cut -c1-175
done
