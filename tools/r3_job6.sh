#!/bin/bash
# timeline of the persistent waves of k_pbl_loop (library built with -DFPX_LANE_STATS)
mkdir -p gpurun_out
for n in 100000000 12500000; do
  timeout -k 10 300 python bench.py --particles $n --warmup 1 --steps 2 --no-pmc --no-cpu-baseline > gpurun_out/r3_timeline_$n.json 2> gpurun_out/r3_timeline_$n.err
  echo "n=$n rc=$?"
done
