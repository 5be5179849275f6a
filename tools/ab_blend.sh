#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blended or fp64_matches_oracle or f32_matches" > gpurun_out/r3_blend_test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3_blend_test.log
[ $rc -eq 0 ] || exit 1
for v in 0 1; do
  for spec in "2 1e8" "2 1e7" "3 1e8"; do
    set -- $spec
    FPX_BLEND_MIN=$v timeout -k 10 400 python bench.py --config $1 --particles $2 --no-cpu-baseline --no-pmc > gpurun_out/ab_blend_${v}_$1_$2.json 2> gpurun_out/ab_blend.err || { echo failed; tail -3 gpurun_out/ab_blend.err; exit 1; }
    echo "blend_min=$v config $1 n=$2 $(grep -o '"k_prep": [0-9.]*' gpurun_out/ab_blend_${v}_$1_$2.json) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_blend_${v}_$1_$2.json)"
  done
done
