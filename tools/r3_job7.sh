#!/bin/bash
# parity suite + config-3 bench (with counter passes) at 1e8 and at the shard of an eighth
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest7.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_gputest7.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench7_c3.json 2> gpurun_out/r3_bench7_c3.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --particles 12500000 --no-cpu-baseline > gpurun_out/r3_bench7_c3_shard.json 2> gpurun_out/r3_bench7_c3_shard.err; echo "bench shard rc=$?"
