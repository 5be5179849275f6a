#!/bin/bash
# round 4 job 37: polar k_prep, this tree against the tree of commit 82c6de6 (worktree _abtmp, built beforehand) on ONE box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r4_j37_*.json
for rep in 1 2; do
for T in new old; do
  D="$GRAFT_REPO_ROOT"; [ $T = old ] && D="$GRAFT_REPO_ROOT/_abtmp"
  for C in "2 --poles --steps 20 --warmup 5" "2 --steps 20 --warmup 5"; do
    N=$(echo "$C" | sed -e 's/[^A-Za-z0-9]//g')_${T}_$rep
    ( cd $D && timeout -k 10 300 python bench.py --config $C --no-cpu-baseline --no-pmc ) > gpurun_out/r4_j37_$N.json 2> gpurun_out/r4_j37.err || { echo "FAILED $T $C"; tail -5 gpurun_out/r4_j37.err; exit 1; }
  done
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j37_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"].get("step_kernels_ms",{})
    print(f.split("j37_")[1], "%.4e"%d["value"], "%.3f ms"%d["ms_per_step"], {n:round(v,3) for n,v in k.items()})
PY
