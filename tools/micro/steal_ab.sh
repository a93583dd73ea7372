mkdir -p gpurun_out/r03w
for lib in steal32 steal8; do
  echo "== $lib"
  NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so timeout -k 10 300 python tools/ablate.py csp 400 100000000 10 2 2>&1 | grep -v amdgpu.ids | cut -c1-280
done
python tools/ablate.py matrix --libs default,nosteal --run "csp 400 100000000 10 2" --run "split 800 100000000 1 2" --run "csp 4000 1000000 10 2" --run "scatter 400 20000000 1 2" 2>&1 | cut -c1-215
