#!/bin/bash
# round 4 job 33: smaller claims in the tail of the Langevin work list (pbl_tail_chunk)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4_gputest33.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest33.log
rm -f gpurun_out/r4_j33_*.json
SH="--config 3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4 --no-cpu-baseline --no-pmc"
for B in 0 8 0 8 1 4 16 32; do
  for k in 1 2 3 4 5 6 7 8 9; do [ -e gpurun_out/r4_j33_shard_t${B}_$k.json ] || break; done
  timeout -k 10 300 python bench.py $SH --opt pbl_tail_chunk=$B > gpurun_out/r4_j33_shard_t${B}_$k.json 2> gpurun_out/r4_j33.err || { echo "FAILED $B"; tail -5 gpurun_out/r4_j33.err; exit 1; }
done
for B in 0 8; do
  timeout -k 10 300 python bench.py --config 3 --steps 5 --warmup 3 --no-cpu-baseline --no-pmc --opt pbl_tail_chunk=$B > gpurun_out/r4_j33_1e8_t${B}.json 2> gpurun_out/r4_j33.err || { echo "FAILED 1e8 $B"; tail -5 gpurun_out/r4_j33.err; exit 1; }
  timeout -k 10 300 python bench.py --config 5 --real 4 --steps 5 --warmup 3 --no-cpu-baseline --no-pmc --opt pbl_tail_chunk=$B > gpurun_out/r4_j33_c5_t${B}.json 2> gpurun_out/r4_j33.err || { echo "FAILED c5 $B"; tail -5 gpurun_out/r4_j33.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j33_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j33_")[1], "%.4e"%d["value"], "%.2f ms"%d["ms_per_step"], {n:round(v,2) for n,v in k.items()})
PY
