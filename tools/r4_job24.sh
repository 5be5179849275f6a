#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for S in 0 3; do
  timeout -k 10 600 python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --steps 8 --warmup 3 --opt pbl_cost_buckets=$S > gpurun_out/r4_j24_shard_b$S.json 2> gpurun_out/r4_j24_shard_b$S.err; echo "shard b$S rc=$?"
done
python - <<'PY'
import json
for S in (0, 3):
    d = json.load(open(f"gpurun_out/r4_j24_shard_b{S}.json")); r = d["roofline"]; v = r["valu"]
    print(S, "%.4e" % d["value"], d["ms_per_step"], r["step_kernels_ms"]["k_pbl_loop"], "traffic GB %.2f" % (r["traffic"] / 1e9), r["traffic_raw"], v.get("insts_valu_per_particle_step"), v.get("lane_utilisation"), v.get("frac_of_launch"))
PY
