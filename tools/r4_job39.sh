#!/bin/bash
# round 4 job 39: the leading aggregate parameters (View, GridP, Parts, PblRec, SeqRng) read through the kernel-argument segment in k_prep, k_pbl_loop, k_pbl_finish (the View alone in k_conccalc, k_wetdepo, k_bkdep): full GPU suite + bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest39.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest39.log
rm -f gpurun_out/r4_j39_*.json
for C in "2 --poles --steps 20 --warmup 5" "2 --steps 20 --warmup 5" "3 --steps 5 --warmup 3" "3 --steps 20 --warmup 5" "5 --real 4 --steps 5 --warmup 3" "3 --real 4 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 20 --warmup 5" "4 --steps 5 --warmup 3"; do
  T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')
  timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j39_$T.json 2> gpurun_out/r4_j39.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j39.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j39_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{}); k2=d.get("kernels_ms") or {}
    print(f.split("j39_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()}, {n:round(v,2) for n,v in (d["roofline"].get("other_kernels_ms") or {}).items()})
PY
