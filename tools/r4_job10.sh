#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "aerosol or three_species or multi_release or nest or golden or f32 or time_slices or domainfill" > gpurun_out/r4_gputest10.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest10.log
timeout -k 10 400 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j10_c5.json 2> gpurun_out/r4_j10_c5.err; echo "c5 rc=$?"
timeout -k 10 400 python bench.py --config 3 --real 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j10_c3f32.json 2> gpurun_out/r4_j10_c3f32.err; echo "c3f32 rc=$?"
python - <<'PY'
import json, glob
for f in ["gpurun_out/r4_j10_c5.json", "gpurun_out/r4_j10_c3f32.json"]:
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
