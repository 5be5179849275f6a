#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py tests/test_checkpoint.py tests/test_concoutput.py -m gpu -x -q -k "not bench_two" > gpurun_out/r4_gputest22.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest22.log
for C in "4" "5 --real 4"; do
  T=$(echo $C | tr -d ' -')
  rm -rf gpurun_out/r4_j22_trace_$T
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_j22_trace_$T -- python $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --no-pmc --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r4_j22_$T.json 2>/dev/null); echo "$C rc=$?"
done
python - <<'PY'
import csv, glob, json
for t in ("4", "5real4"):
    d = json.load(open(f"gpurun_out/r4_j22_{t}.json")); print(t, "%.4e" % d["value"], d["ms_per_step"])
    f = glob.glob(f"gpurun_out/r4_j22_trace_{t}/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_wetdepo", "k_conccalc")):
            print("   ", r["Name"][:50], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
