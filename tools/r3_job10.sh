#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --particles 12500000 --no-cpu-baseline > gpurun_out/r3_bench10_shard.json 2> gpurun_out/r3_bench10_shard.err; echo "bench shard rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r3_bench10_c3.json 2> gpurun_out/r3_bench10_c3.err; echo "bench c3 rc=$?"
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench10_shard.json gpurun_out/r3_bench10_c3.json
grep -h -o '"lane_utilisation": [0-9.]*' gpurun_out/r3_bench10_shard.json gpurun_out/r3_bench10_c3.json
