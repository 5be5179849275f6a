#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc"
for S in 0 3; do
  timeout -k 10 300 python $B --steps 8 --warmup 3 --opt pbl_cost_buckets=$S > gpurun_out/r4_j17_b$S.json 2> gpurun_out/r4_j17_b$S.err; echo "b$S rc=$?"
done
export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_lanestats.so
for S in 0 3; do
  timeout -k 10 300 python $B --steps 4 --warmup 2 --opt pbl_cost_buckets=$S > gpurun_out/r4_j17_lanes_b$S.json 2> gpurun_out/r4_j17_lanes_b$S.err; echo "lanes b$S rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j17_b*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
for f in sorted(glob.glob("gpurun_out/r4_j17_lanes_*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["config"]["options"])
    for k, v in r.get("lane_stats", {}).items():
        print("   ", k, v)
PY
