#!/bin/bash
# round 3, first GPU trip: the whole GPU suite, the default bench line, and the self-launched two-rank rehearsal
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_gputest1.log
tail -5 gpurun_out/r3_gputest1.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_c3_a.json 2> gpurun_out/r3_bench_c3_a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r3_bench_c3_a.json
timeout -k 10 300 python bench.py --gpus 2 --particles 2e7 --no-cpu-baseline > gpurun_out/r3_bench_n2_rehearsal.json 2> gpurun_out/r3_bench_n2_rehearsal.err; echo "n2 rc=$?"
tail -c 800 gpurun_out/r3_bench_n2_rehearsal.json
