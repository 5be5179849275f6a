#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for SI in 4 8 16; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 --sort-interval $SI > gpurun_out/r4_j27_1e8_si$SI.json 2> gpurun_out/r4_j27_1e8_si$SI.err; echo "1e8 si$SI rc=$?"
  timeout -k 10 400 python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 20 --warmup 5 --sort-interval $SI > gpurun_out/r4_j27_shard_si$SI.json 2> gpurun_out/r4_j27_shard_si$SI.err; echo "shard si$SI rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j27_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.4e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
