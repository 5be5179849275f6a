#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "time_slices or golden or f32 or blend or counter" > gpurun_out/r4_gputest19.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest19.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j19_1e8.json 2> gpurun_out/r4_j19_1e8.err; echo "1e8 rc=$?"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 --opt pbl_slices=0,0 --opt pbl_drain_lanes=0 > gpurun_out/r4_j19_1e8_susp.json 2> gpurun_out/r4_j19_1e8_susp.err; echo "1e8 susp rc=$?"
timeout -k 10 300 python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > gpurun_out/r4_j19_shard.json 2> gpurun_out/r4_j19_shard.err; echo "shard rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j19_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
