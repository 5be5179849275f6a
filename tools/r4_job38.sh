#!/bin/bash
# round 4 job 38: the View read through the kernel-argument segment in k_prep / k_pbl_finish (default build) and also in k_pbl_loop (variant build)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest38.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest38.log
rm -f gpurun_out/r4_j38_*.json
for L in default loopkarg; do
  if [ $L = default ]; then unset FPX_LIBRARY; else export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$L.so; fi
  for C in "2 --poles --steps 20 --warmup 5" "2 --steps 20 --warmup 5" "3 --steps 5 --warmup 3" "3 --poles --steps 5 --warmup 3" "5 --real 4 --steps 5 --warmup 3" "3 --real 4 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4" "4 --steps 5 --warmup 3"; do
    [ $L = loopkarg ] && case "$C" in 2*) continue;; esac
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_$L
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j38_$T.json 2> gpurun_out/r4_j38.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j38.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j38_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j38_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
