#!/bin/bash
# round 4, GPU job 1: the whole GPU suite on the time-slice code, then the slice schedules at the shard of an eighth
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest1.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest1.log
B="python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
i=0
for S in "0" "64,64,64,0" "32,64,128,0" "48,48,48,48,48,48,0" "96,96,0" "24,48,96,192,0"; do
  i=$((i+1))
  timeout -k 10 300 $B --opt pbl_slices=$S > gpurun_out/r4_j1_shard_$i.json 2> gpurun_out/r4_j1_shard_$i.err; echo "shard $S rc=$?"
done
timeout -k 10 300 $B --steps 3 --warmup 2 --opt pbl_slices=32,32,32,32,32,32,32,32,32,32,32,32,32,32,0 --opt verbose=2 > gpurun_out/r4_j1_lists.json 2> gpurun_out/r4_j1_lists.err; echo "lists rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j1_shard_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], r["step_kernels_ms"], d["config"]["time_blended_packs"])
    except Exception as e:
        print(f, "failed", e)
PY
grep "Langevin lists" gpurun_out/r4_j1_lists.err | tail -3
