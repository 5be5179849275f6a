#!/bin/bash
# round 4 job 36: gas instances of k_prep / k_pbl_finish without the settling code; species table in device memory; full GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest36.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r4_gputest36.log
rm -f gpurun_out/r4_j36_*.json
for C in "2 --poles --steps 20 --warmup 5" "2 --steps 20 --warmup 5" "3 --poles --steps 5 --warmup 3" "3 --steps 5 --warmup 3" "5 --real 4 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4"; do
  for O in 0 1; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_init$O
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc --opt prep_init_always=$O > gpurun_out/r4_j36_$T.json 2> gpurun_out/r4_j36.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j36.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j36_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j36_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
