#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest11.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_gputest11.log
timeout -k 10 500 python bench.py --config 4 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench11_c4.json 2> gpurun_out/r3_bench11_c4.err; echo "bench c4 rc=$?"
timeout -k 10 500 python bench.py --config 5 --real 4 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench11_c5.json 2> gpurun_out/r3_bench11_c5.err; echo "bench c5 rc=$?"
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r3_bench11_c4.json gpurun_out/r3_bench11_c5.json
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench11_c4.json gpurun_out/r3_bench11_c5.json
timeout -k 10 500 python bench.py --config 3 --real 4 --no-cpu-baseline --no-pmc > gpurun_out/r3_bench11_c3f.json 2> gpurun_out/r3_bench11_c3f.err; echo "bench c3f rc=$?"
grep -h -o '"step_kernels_ms": {[^}]*}' gpurun_out/r3_bench11_c3f.json
