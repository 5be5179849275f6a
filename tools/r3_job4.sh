#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_$1.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3_gputest_$1.log
[ $rc -eq 0 ] || exit 1
for c in "5 --real 4" "4"; do
  timeout -k 10 400 python bench.py --config $c --no-cpu-baseline --no-pmc > gpurun_out/r3_tmp.json 2> gpurun_out/r3_tmp.err; python3 -c "
import json; d=json.load(open('gpurun_out/r3_tmp.json')); print('config $c', 'value %.4g ms/step %.1f' % (d['value'], d['ms_per_step']), d['roofline']['step_kernels_ms'])"
done
