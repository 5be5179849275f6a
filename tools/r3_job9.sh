#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_gputest9.log
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench9_c2_1.json 2> gpurun_out/r3_bench9_c2.err; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --config 2 --particles 1e8 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench9_c2_1e8.json 2> gpurun_out/r3_bench9_c2_1e8.err; echo "bench c2 1e8 rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-pmc > gpurun_out/r3_bench9_c3.json 2> gpurun_out/r3_bench9_c3.err; echo "bench c3 rc=$?"
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench9_c2_1.json gpurun_out/r3_bench9_c2_1e8.json gpurun_out/r3_bench9_c3.json
