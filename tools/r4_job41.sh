#!/bin/bash
# round 4 job 41: fp64 k_prep instances with a nest table / dry deposition at the three-wave budget (default build) against two waves (variant prep2w), one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_release.py -m gpu -x -q > gpurun_out/r4_gputest41.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest41.log
rm -f gpurun_out/r4_j41_*.json
for rep in 1 2; do
for L in default prep2w; do
  if [ $L = default ]; then unset FPX_LIBRARY; else export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$L.so; fi
  for C in "5 --real 8 --particles 30000000 --steps 5 --warmup 3" "5 --real 8 --without aerosol --particles 30000000 --steps 5 --warmup 3" "5 --real 8 --without nest --particles 30000000 --steps 5 --warmup 3"; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_${L}_$rep
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j41_$T.json 2> gpurun_out/r4_j41.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j41.err; exit 1; }
  done
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j41_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j41_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
