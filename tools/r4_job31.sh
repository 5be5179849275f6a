#!/bin/bash
# round 4 job 31: cost estimate of the work list from the passes of the particle's last boundary-layer step (pbl_cost_buckets +4)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_release.py tests/test_checkpoint.py -m gpu -x -q > gpurun_out/r4_gputest31.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest31.log
rm -f gpurun_out/r4_j31_*.json
SH="--config 3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4 --no-cpu-baseline --no-pmc"
for B in 3 7 3 7 5 6; do
  for k in 1 2 3 4 5 6 7 8 9; do [ -e gpurun_out/r4_j31_shard_b${B}_$k.json ] || break; done
  timeout -k 10 300 python bench.py $SH --opt pbl_cost_buckets=$B > gpurun_out/r4_j31_shard_b${B}_$k.json 2> gpurun_out/r4_j31.err || { echo "FAILED $B"; tail -5 gpurun_out/r4_j31.err; exit 1; }
done
for B in 0 7 3; do
  timeout -k 10 300 python bench.py --config 3 --steps 5 --warmup 3 --no-cpu-baseline --no-pmc --opt pbl_cost_buckets=$B > gpurun_out/r4_j31_1e8_b${B}.json 2> gpurun_out/r4_j31.err || { echo "FAILED 1e8 $B"; tail -5 gpurun_out/r4_j31.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j31_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j31_")[1], "%.4e"%d["value"], "%.2f ms"%d["ms_per_step"], {n:round(v,2) for n,v in k.items()})
PY
