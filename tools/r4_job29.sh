#!/bin/bash
# round 4 job 29: the pick() fix (kernel argument out of scratch in the f32 k_prep instances with initialize()); full GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest29.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest29.log
rm -f gpurun_out/r4_j29_*.json
for C in "3 --real 4 --steps 5 --warmup 3" "3 --real 4 --steps 5 --warmup 3 --opt prep_init_always=1" "5 --real 4 --steps 5 --warmup 3" "5 --real 4 --steps 5 --warmup 3 --opt prep_init_always=1" "3 --steps 5 --warmup 3" "3 --steps 5 --warmup 3 --opt prep_init_always=1" "4 --steps 5 --warmup 3"; do
  T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')
  timeout -k 10 300 python bench.py --config $C --no-cpu-baseline > gpurun_out/r4_j29_$T.json 2> gpurun_out/r4_j29_$T.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j29_$T.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j29_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get("kernels_ms",{})
        print(f.split("j29_")[1], "%.4e"%d["value"], "%.2f ms"%d["ms_per_step"], {n:round(v,2) for n,v in k.items() if n in("k_prep","k_pbl_loop","k_pbl_finish","k_conccalc","k_wetdepo")})
    except Exception as e: print(f, "ERR", e)
PY
