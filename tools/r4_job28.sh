#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rm -f gpurun_out/r4_j28_*.json
for C in "2 --steps 20 --warmup 5" "2 --steps 20 --warmup 5 --opt prep_init_always=1" "3 --real 4 --steps 5 --warmup 3" "3 --real 4 --steps 5 --warmup 3 --opt prep_init_always=1" "5 --real 4 --without nest --steps 5 --warmup 3 --opt prep_init_always=1" "5 --real 4 --without aerosol --steps 5 --warmup 3 --opt prep_init_always=1"; do
  T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')
  timeout -k 10 400 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j28_$T.json 2> gpurun_out/r4_j28_$T.err; echo "$C rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j28_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.4e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
