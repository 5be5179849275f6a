#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for W in "nest" "aerosol" "nest,aerosol" "wet"; do
  T=$(echo $W | tr ',' '_')
  timeout -k 10 400 python bench.py --config 5 --real 4 --without $W --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j11_c5_no_$T.json 2> gpurun_out/r4_j11_c5_no_$T.err; echo "c5 without $W rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j11_c5_no_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
