#!/bin/bash
# profiles of the round: kernel stats + SQ / FETCH / WRITE counter passes (tools/collect_profile.sh) and the default bench line
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "$@"; do
  case "$spec" in
    c3)   TAG=c3_1e+08; ARGS="--config 3";;
    c2a)  TAG=c2_1e+07; ARGS="--config 2";;
    c2b)  TAG=c2_1e+08; ARGS="--config 2 --particles 1e8";;
    c4)   TAG=c4_1e+08; ARGS="--config 4";;
    c5)   TAG=c5_1e+08; ARGS="--config 5 --real 4";;
    c3f)  TAG=c3f32_1e+08; ARGS="--config 3 --real 4";;
  esac
  timeout -k 10 500 python bench.py $ARGS > gpurun_out/r3_${TAG}_bench_default.json 2> gpurun_out/r3_${TAG}_bench_default.err; echo "$TAG bench rc=$?"
  bash tools/collect_profile.sh $TAG $ARGS --steps 4 --warmup 2 || echo "$TAG profile failed"
done
