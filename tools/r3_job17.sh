#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 100000000 12500000; do
  timeout -k 10 400 python bench.py --particles $n --no-cpu-baseline > gpurun_out/r3_bench19_$n.json 2> gpurun_out/r3_bench19_$n.err || { echo "bench failed $n"; tail -3 gpurun_out/r3_bench19_$n.err; exit 1; }
done
grep -h -o '"ms_per_step": [0-9.]*\|"k_pbl_loop": [0-9.]*\|"lane_utilisation": [0-9.]*\|"insts_valu_per_particle_step": [0-9.]*' gpurun_out/r3_bench19_100000000.json gpurun_out/r3_bench19_12500000.json
