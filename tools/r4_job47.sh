#!/bin/bash
# round 4 job 47: instruction-level work on the fine loop (Philox products, native abs / max / min / sign, sqrt without selects): parity tests + bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest47.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest47.log
rm -f gpurun_out/r4_j47_*.json
for C in "3 --steps 5 --warmup 3" "3 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4" "5 --real 4 --steps 5 --warmup 3" "3 --real 4 --steps 5 --warmup 3" "2 --steps 20 --warmup 5"; do
  T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')
  for k in 1 2 3; do [ -e gpurun_out/r4_j47_${T}_$k.json ] || break; done
  timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j47_${T}_$k.json 2> gpurun_out/r4_j47.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j47.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j47_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j47_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
