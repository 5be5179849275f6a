#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --config 3 --real 4 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench12_c3f.json 2> gpurun_out/r3_bench12_c3f.err; echo "bench c3f rc=$?"
timeout -k 10 500 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench12_c5.json 2> gpurun_out/r3_bench12_c5.err; echo "bench c5 rc=$?"
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r3_bench12_c3f.json gpurun_out/r3_bench12_c5.json
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench12_c3f.json gpurun_out/r3_bench12_c5.json
