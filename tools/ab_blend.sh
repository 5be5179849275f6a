#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_checkpoint.py -m gpu -x -q > gpurun_out/r3_lut_test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_lut_test.log
[ $rc -eq 0 ] || exit 1
for spec in "3 1e8 8" "3 1e8 8" "3 12500000 8" "3 1e8 4"; do
  set -- $spec
  timeout -k 10 400 python bench.py --config $1 --particles $2 --real $3 --no-cpu-baseline --no-pmc > gpurun_out/ab_$1_$2_$3.json 2> gpurun_out/ab_blend.err || { echo failed; tail -3 gpurun_out/ab_blend.err; exit 1; }
  echo "config $1 n=$2 real $3 $(grep -o '"k_pbl_loop": [0-9.]*' gpurun_out/ab_$1_$2_$3.json) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$1_$2_$3.json)"
done
