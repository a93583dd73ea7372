#!/bin/bash
# what the board draws and the clocks it reports while bench.py runs (rocm-smi polled beside it; reading only)
out=gpurun_out/power; mkdir -p $out
rocm-smi --showmaxpower --showpower --showclocks > $out/idle.txt 2>&1
timeout -k 10 200 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-lazy-leg > $out/bench.json 2> $out/bench.err &
pid=$!
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" >> $out/busy.txt
  echo "--" >> $out/busy.txt
  sleep 0.5
done
wait $pid
grep -i "max\|power" $out/idle.txt | head -8
grep -i 'power' $out/busy.txt | awk '{print $NF}' | tr '\n' ' '; echo; grep -i 'sclk' $out/busy.txt | awk '{print $NF}' | tr '\n' ' '; echo
