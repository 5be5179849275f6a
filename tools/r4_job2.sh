#!/bin/bash
# round 4, GPU job 2: stability classes as concurrent launch sequences + time slices, at the shard of an eighth
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "time_slices or counter_rng or sort" > gpurun_out/r4_gputest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest2.log
B="python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
i=0
for S in "0:0" "0:1" "64,64,64,0:1" "32,32,32,32,32,32,32,32,0:1" "96,96,0:1" "48,96,0:1" "128,0:1" "16,16,16,16,16,16,16,16,16,16,16,16,16,16,0:1"; do
  i=$((i+1))
  timeout -k 10 300 $B --opt pbl_slices=${S%%:*} --opt pbl_class_streams=${S##*:} > gpurun_out/r4_j2_shard_$i.json 2> gpurun_out/r4_j2_shard_$i.err; echo "shard $S rc=$?"
done
timeout -k 10 300 $B --steps 3 --warmup 2 --opt pbl_slices=32,32,32,32,32,32,32,32,32,32,32,32,32,32,0 --opt verbose=2 > gpurun_out/r4_j2_lists.json 2> gpurun_out/r4_j2_lists.err; echo "lists rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j2_shard_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
grep "Langevin lists" gpurun_out/r4_j2_lists.err | tail -4
