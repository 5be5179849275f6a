#!/bin/bash
# round 4 job 48: a variant build against the default on one box (config 3 at 1e8 and at the shard, config 5): FPX_LIBRARY=libflexpart_amd_$1.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$1
rm -f gpurun_out/r4_j48_*.json
for rep in 1 2; do
for L in default $V; do
  if [ $L = default ]; then unset FPX_LIBRARY; else export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$L.so; fi
  for C in "3 --steps 5 --warmup 3" "3 --particles 12500000 --global-particles 100000000 --steps 8 --warmup 4" "5 --real 4 --steps 5 --warmup 3"; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_${L}_$rep
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc > gpurun_out/r4_j48_$T.json 2> gpurun_out/r4_j48.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j48.err; exit 1; }
  done
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j48_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j48_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
