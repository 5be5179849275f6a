#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_checkpoint.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/r3_gputest16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_gputest16.log
[ $rc -eq 0 ] || exit 1
for n in 100000000 12500000; do
  timeout -k 10 400 python bench.py --particles $n --no-cpu-baseline > gpurun_out/r3_bench16_$n.json 2> gpurun_out/r3_bench16_$n.err || { echo "bench failed $n"; tail -3 gpurun_out/r3_bench16_$n.err; exit 1; }
done
grep -h -o '"ms_per_step": [0-9.]*\|"k_pbl_loop": [0-9.]*\|"lane_utilisation": [0-9.]*\|"insts_valu_per_particle_step": [0-9.]*' gpurun_out/r3_bench16_100000000.json gpurun_out/r3_bench16_12500000.json
