#!/bin/bash
# A/B on ONE box: round 3's tree against the current one, config 3 at 1e8 and at the shard of an eighth.
# Needs the round-3 tree next to this one first:  git worktree add -f _r3tmp 0e65107 && (cd _r3tmp && python -c "import __graft_entry__ as g; g.build_library()")
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
  (cd _r3tmp && FPX_BLEND_MIN=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > ../gpurun_out/r4_j20_r3_1e8_$i.json 2> ../gpurun_out/r4_j20_r3_1e8_$i.err); echo "r3 1e8 rc=$?"
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j20_r4_1e8_$i.json 2> gpurun_out/r4_j20_r4_1e8_$i.err; echo "r4 1e8 rc=$?"
done
(cd _r3tmp && FPX_BLEND_MIN=1 timeout -k 10 400 python bench.py --particles 12500000 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > ../gpurun_out/r4_j20_r3_shard.json 2> ../gpurun_out/r4_j20_r3_shard.err); echo "r3 shard rc=$?"
timeout -k 10 400 python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > gpurun_out/r4_j20_r4_shard.json 2> gpurun_out/r4_j20_r4_shard.err; echo "r4 shard rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j20_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
