#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py tests/test_checkpoint.py tests/test_concoutput.py -m gpu -x -q -k "not bench_two" > gpurun_out/r4_gputest16.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest16.log
timeout -k 10 400 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j16_c5.json 2> gpurun_out/r4_j16_c5.err; echo "c5 rc=$?"
timeout -k 10 400 python bench.py --config 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j16_c4.json 2> gpurun_out/r4_j16_c4.err; echo "c4 rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_j16_trace -- python $GRAFT_REPO_ROOT/bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc --steps 4 --warmup 2 > /dev/null 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, glob, csv
for f in ["gpurun_out/r4_j16_c5.json", "gpurun_out/r4_j16_c4.json"]:
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
f = max(glob.glob("gpurun_out/r4_j16_trace/**/*kernel_stats.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_wetdepo", "k_conccalc", "k_prep", "k_pbl_finish", "k_pbl_loop")):
        print(r["Name"][:60], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
