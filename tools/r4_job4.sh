#!/bin/bash
# round 4, GPU job 4: class-pure waves + drain re-packing, at the shard of an eighth and at 1e8
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "time_slices or counter_rng or sort or golden" > gpurun_out/r4_gputest4.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest4.log
B="python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
i=0
for S in "0:0" "0,0,0,0:32" "0,0,0,0:48" "0,0,0,0:16" "0,0:32" "0,0,0,0,0,0:40" "0,0,0,0,0,0:56" "0,0,0:24"; do
  i=$((i+1))
  timeout -k 10 300 $B --opt pbl_slices=${S%%:*} --opt pbl_drain_lanes=${S##*:} > gpurun_out/r4_j4_shard_$i.json 2> gpurun_out/r4_j4_shard_$i.err; echo "shard $S rc=$?"
done
timeout -k 10 300 $B --steps 3 --warmup 2 --opt verbose=2 > gpurun_out/r4_j4_lists.json 2> gpurun_out/r4_j4_lists.err; echo "lists rc=$?"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 --opt pbl_slices=0 > gpurun_out/r4_j4_1e8_single.json 2> gpurun_out/r4_j4_1e8_single.err; echo "1e8 single rc=$?"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j4_1e8_default.json 2> gpurun_out/r4_j4_1e8_default.err; echo "1e8 default rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j4_shard_*.json")) + ["gpurun_out/r4_j4_1e8_single.json", "gpurun_out/r4_j4_1e8_default.json"]:
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
grep "Langevin lists" gpurun_out/r4_j4_lists.err | tail -2
