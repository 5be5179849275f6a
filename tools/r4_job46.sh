#!/bin/bash
# round 4 job 46: k_prep at the four-wave register budget (variant prep4w: 128 VGPRs, 79 spilled) against three waves, one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r4_j46_*.json
for L in default prep4w; do
  if [ $L = default ]; then unset FPX_LIBRARY; else export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$L.so; fi
  for C in "2 --steps 20 --warmup 5" "2 --particles 100000000 --steps 8 --warmup 3" "2 --poles --steps 20 --warmup 5"; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_$L
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j46_$T.json 2> gpurun_out/r4_j46.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j46.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j46_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j46_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
