#!/bin/bash
# round 4: the evidence files of profiles/r4 on the final code (run through gpurun; tools/summarize_profile.py afterwards)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
case "$1" in
 a)
  timeout -k 10 900 python bench.py > gpurun_out/r4_c3_1e+08_bench_default.json 2> gpurun_out/r4_c3_default.err; echo "c3 default rc=$?"
  bash tools/collect_profile.sh c3_1e+08 --config 3 --steps 4 --warmup 2; echo "c3 rc=$?"
  bash tools/collect_profile.sh c3_1.25e+07 --config 3 --particles 12500000 --global-particles 1e8 --steps 8 --warmup 3; echo "c3 shard rc=$?"
  ;;
 b)
  timeout -k 10 900 python bench.py --config 5 --real 4 > gpurun_out/r4_c5_1e+08_bench_default.json 2> gpurun_out/r4_c5_default.err; echo "c5 default rc=$?"
  bash tools/collect_profile.sh c5_1e+08 --config 5 --real 4 --steps 4 --warmup 2; echo "c5 rc=$?"
  bash tools/collect_profile.sh c4_1e+08 --config 4 --steps 4 --warmup 2; echo "c4 rc=$?"
  ;;
 c)
  timeout -k 10 600 python bench.py --config 2 > gpurun_out/r4_c2_1e+07_bench_default.json 2> gpurun_out/r4_c2_default.err; echo "c2 default rc=$?"
  bash tools/collect_profile.sh c2_1e+07 --config 2 --steps 20 --warmup 5; echo "c2 rc=$?"
  bash tools/collect_profile.sh c2p_1e+07 --config 2 --poles --steps 20 --warmup 5; echo "c2p rc=$?"
  bash tools/collect_profile.sh c2_1e+08 --config 2 --particles 1e8 --steps 8 --warmup 3; echo "c2 1e8 rc=$?"
  bash tools/collect_profile.sh c3p_1e+08 --config 3 --poles --steps 4 --warmup 2; echo "c3p rc=$?"
  timeout -k 10 600 python bench.py --config 3 --real 4 > gpurun_out/r4_c3f32_1e+08_bench_default.json 2> gpurun_out/r4_c3f32_default.err; echo "c3f32 default rc=$?"
  ;;
esac
