#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r3_bench13_c3.json 2> gpurun_out/r3_bench13_c3.err; echo "bench c3 rc=$?"
timeout -k 10 300 python bench.py --particles 12500000 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench13_shard.json 2> gpurun_out/r3_bench13_shard.err; echo "bench shard rc=$?"
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench13_c3.json gpurun_out/r3_bench13_shard.json
grep -h -o '"frac_of_launch": [0-9.]*' gpurun_out/r3_bench13_c3.json
