#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_checkpoint.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/r3_cls_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3_cls_parity.log
[ $rc -eq 0 ] || exit 1
for n in 1e8 12500000 2e7; do
  timeout -k 10 300 python bench.py --particles $n --no-cpu-baseline > gpurun_out/r3_cls_$n.json 2> gpurun_out/r3_cls.err || { echo "bench failed n=$n"; tail -3 gpurun_out/r3_cls.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3_cls_$n.json')); v=d['roofline'].get('valu',{}); print('n $n value %.4g loop %.2f ms' % (d['value'], d['roofline']['step_kernels_ms']['k_pbl_loop']), {k: v.get(k) for k in ('insts_valu_per_particle_step','lane_utilisation','frac_of_launch')})"
done
