#!/bin/bash
# round 4 job 40: k_pbl_loop with all leading aggregates through the kernel-argument segment (default build) against the View alone (variant looponlyview), one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r4_j40_*.json
for rep in 1 2; do
for L in default looponlyview; do
  if [ $L = default ]; then unset FPX_LIBRARY; else export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$L.so; fi
  for C in "3 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4" "5 --real 4 --steps 5 --warmup 3"; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_${L}_$rep
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j40_$T.json 2> gpurun_out/r4_j40.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j40.err; exit 1; }
  done
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j40_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j40_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
