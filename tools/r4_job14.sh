#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_partoutput.py tests/test_verttransform.py -m gpu -x -q -k "polar or golden or baseline or global" > gpurun_out/r4_gputest14.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest14.log
for C in "2 --steps 20 --warmup 5" "2 --poles --steps 20 --warmup 5" "3 --poles --steps 5 --warmup 3"; do
  T=$(echo $C | tr -d ' -')
  timeout -k 10 400 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j14_$T.json 2> gpurun_out/r4_j14_$T.err; echo "$C rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j14_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
