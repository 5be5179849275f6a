#!/bin/bash
# round 4 job 35: species table in device memory (spec_row) instead of select chains into the kernel argument; LDS-typed queue pointer in k_pbl_finish
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_release.py -m gpu -x -q > gpurun_out/r4_gputest35.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest35.log
rm -f gpurun_out/r4_j35_*.json
for C in "2 --poles --steps 20 --warmup 5" "2 --steps 20 --warmup 5" "3 --poles --steps 5 --warmup 3" "5 --real 4 --steps 5 --warmup 3" "5 --real 8 --particles 30000000 --steps 5 --warmup 3" "5 --real 4 --poles --particles 30000000 --steps 5 --warmup 3"; do
  for O in 0 1; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_init$O
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc --opt prep_init_always=$O > gpurun_out/r4_j35_$T.json 2> gpurun_out/r4_j35.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j35.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j35_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j35_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
