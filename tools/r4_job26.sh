#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_checkpoint.py -m gpu -x -q > gpurun_out/r4_gputest26.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest26.log
timeout -k 10 300 python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > gpurun_out/r4_j26_shard.json 2> gpurun_out/r4_j26_shard.err; echo "shard rc=$?"
timeout -k 10 400 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j26_c5.json 2> gpurun_out/r4_j26_c5.err; echo "c5 rc=$?"
timeout -k 10 400 python bench.py --config 5 --real 4 --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > gpurun_out/r4_j26_c5_shard.json 2> gpurun_out/r4_j26_c5_shard.err; echo "c5 shard rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j26_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.4e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
