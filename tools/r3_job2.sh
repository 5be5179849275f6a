#!/bin/bash
# parity of the step kernels + the default bench line (no CPU baseline)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${1:-x}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_checkpoint.py -m gpu -x -q > gpurun_out/r3_parity_$TAG.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -4 gpurun_out/r3_parity_$TAG.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/r3_bench_$TAG.json 2> gpurun_out/r3_bench_$TAG.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/r3_bench_$TAG.json"))
r=d["roofline"]
print("value %.4g  ms/step %.1f  loop %.1f ms  prep %.2f  finish %.2f" % (d["value"], d["ms_per_step"], r["step_kernels_ms"]["k_pbl_loop"], r["step_kernels_ms"]["k_prep"], r["step_kernels_ms"]["k_pbl_finish"]))
v=r.get("valu",{})
print({k:v.get(k) for k in ("insts_valu_per_particle_step","lane_utilisation","frac_of_launch","cycles_per_valu_inst")})
PY
