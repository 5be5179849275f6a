#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_partoutput.py tests/test_release.py -m gpu -x -q > gpurun_out/r4_gputest18.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest18.log
for C in "2 --steps 20 --warmup 5" "2 --poles --steps 20 --warmup 5" "3 --poles --steps 5 --warmup 3" "3 --steps 5 --warmup 3"; do
  T=$(echo $C | tr -d ' -')
  timeout -k 10 400 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j18_$T.json 2> gpurun_out/r4_j18_$T.err; echo "$C rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j18_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
