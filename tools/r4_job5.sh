#!/bin/bash
# round 4, GPU job 5: cost buckets below the class in the work-list key
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "time_slices or counter_rng or sort or golden" > gpurun_out/r4_gputest5.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest5.log
B="python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
i=0
for S in "0:0" "1:0" "0:0" "1:0" "1:1"; do
  i=$((i+1))
  X=""; if [ "${S##*:}" = "1" ]; then X="--opt pbl_slices=0,0,0,0 --opt pbl_drain_lanes=32"; fi
  timeout -k 10 300 $B --opt pbl_cost_buckets=${S%%:*} $X > gpurun_out/r4_j5_shard_$i.json 2> gpurun_out/r4_j5_shard_$i.err; echo "shard $S rc=$?"
done
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 --opt pbl_cost_buckets=0 > gpurun_out/r4_j5_1e8_b0.json 2> gpurun_out/r4_j5_1e8_b0.err; echo "1e8 b0 rc=$?"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 --opt pbl_cost_buckets=1 > gpurun_out/r4_j5_1e8_b1.json 2> gpurun_out/r4_j5_1e8_b1.err; echo "1e8 b1 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j5_shard_*.json")) + ["gpurun_out/r4_j5_1e8_b0.json", "gpurun_out/r4_j5_1e8_b1.json"]:
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
