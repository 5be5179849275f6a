#!/bin/bash
# round 4, GPU job 9: fast get_settling in the Langevin kernel (config 5), bucket schemes, 2-rank bench test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py -m gpu -x -q -k "aerosol or three_species or multi_release or nest or golden or f32 or bench_two or time_slices" > gpurun_out/r4_gputest9.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest9.log
timeout -k 10 400 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j9_c5.json 2> gpurun_out/r4_j9_c5.err; echo "c5 rc=$?"
B="python bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
for S in 0 1 2 3; do
  timeout -k 10 300 $B --opt pbl_cost_buckets=$S > gpurun_out/r4_j9_shard_b$S.json 2> gpurun_out/r4_j9_shard_b$S.err; echo "shard b$S rc=$?"
done
for S in 2 3; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc --steps 6 --warmup 3 --opt pbl_cost_buckets=$S > gpurun_out/r4_j9_1e8_b$S.json 2> gpurun_out/r4_j9_1e8_b$S.err; echo "1e8 b$S rc=$?"
done
python - <<'PY'
import json, glob
for f in ["gpurun_out/r4_j9_c5.json"] + sorted(glob.glob("gpurun_out/r4_j9_shard_b*.json")) + sorted(glob.glob("gpurun_out/r4_j9_1e8_b*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["config"]["options"], "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
