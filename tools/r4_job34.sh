#!/bin/bash
# round 4 job 34: does the k_prep instance with initialize() cost anything when no particle is new? (prep_init_always=1 against the default, polar / nest / aerosol instances)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r4_j34_*.json
for C in "2 --steps 20 --warmup 5" "2 --poles --steps 20 --warmup 5" "3 --steps 5 --warmup 3" "3 --poles --steps 5 --warmup 3" "5 --real 8 --particles 30000000 --steps 5 --warmup 3" "5 --real 8 --poles --particles 30000000 --steps 5 --warmup 3" "5 --real 4 --poles --particles 30000000 --steps 5 --warmup 3"; do
  for O in 0 1; do
    T=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_init$O
    timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc --opt prep_init_always=$O > gpurun_out/r4_j34_$T.json 2> gpurun_out/r4_j34.err || { echo "FAILED $C"; tail -5 gpurun_out/r4_j34.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j34_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j34_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
