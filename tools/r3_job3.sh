#!/bin/bash
# configs 4 and 5 on current code: bench lines (with cpu_baseline) and rocprofv3 kernel stats + PMC passes
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
which=${1:-4}
if [ "$which" = 4 ]; then ARGS="--config 4"; TAG=c4; else ARGS="--config 5 --real 4"; TAG=c5; fi
timeout -k 10 500 python bench.py $ARGS > gpurun_out/r3_${TAG}_1e8_bench.json 2> gpurun_out/r3_${TAG}_1e8_bench.err; echo "bench 1e8 rc=$?"
timeout -k 10 300 python bench.py $ARGS --particles 12500000 --no-cpu-baseline > gpurun_out/r3_${TAG}_shard_bench.json 2> gpurun_out/r3_${TAG}_shard_bench.err; echo "bench shard rc=$?"
bash tools/collect_profile.sh ${TAG}_1e+08 $ARGS --steps 4 --warmup 2 || echo "profile failed"
python3 - <<PY
import json
for f in ("gpurun_out/r3_${TAG}_1e8_bench.json","gpurun_out/r3_${TAG}_shard_bench.json"):
    d=json.load(open(f)); r=d["roofline"]
    print(f, "value %.4g ms/step %.1f" % (d["value"], d["ms_per_step"]), r["step_kernels_ms"], d.get("cpu_baseline"))
PY
