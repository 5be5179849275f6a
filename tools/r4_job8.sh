#!/bin/bash
# round 4, GPU job 8: the bench line on current code; profiles of config 5 (f32 aerosol instance) and of the full-globe (polar) variants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r4_j8_c3_default.json 2> gpurun_out/r4_j8_c3_default.err; echo "bench default rc=$?"
bash tools/collect_profile.sh c5_1e+08 --config 5 --real 4 --steps 4 --warmup 2; echo "c5 rc=$?"
bash tools/collect_profile.sh c2_1e+07 --config 2 --steps 20 --warmup 5; echo "c2 rc=$?"
bash tools/collect_profile.sh c2p_1e+07 --config 2 --poles --steps 20 --warmup 5; echo "c2p rc=$?"
bash tools/collect_profile.sh c3p_1e+08 --config 3 --poles --steps 4 --warmup 2; echo "c3p rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_j8_c3_default.json"))
print("default", "%.3e" % d["value"], d["ms_per_step"], d["roofline"]["step_kernels_ms"], d["roofline"].get("valu", {}).get("frac_of_launch"), d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for t in ("c5_1e+08", "c2_1e+07", "c2p_1e+07", "c3p_1e+08"):
    try:
        d = json.load(open(f"gpurun_out/prof_{t}/stats/bench.json"))
        print(t, "%.3e" % d["value"], d["ms_per_step"], d["roofline"]["step_kernels_ms"])
    except Exception as e:
        print(t, "failed", e)
PY
