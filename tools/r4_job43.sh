#!/bin/bash
# round 4 job 43: k_conccalc / k_wetdepo with all aggregates read in place: kernel trace of configs 5 (f32) and 4 (fp64), tests of the grids
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/r4_gputest43.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest43.log
for C in 5 4; do
  R=8; [ $C = 5 ] && R=4
  rm -rf gpurun_out/j43_c$C
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/j43_c$C -- python $GRAFT_REPO_ROOT/bench.py --config $C --real $R --steps 5 --warmup 3 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r4_j43_c$C.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4_j43.err ) || { echo FAILED $C; tail -5 gpurun_out/r4_j43.err; exit 1; }
  python - "$C" <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/j43_c{sys.argv[1]}/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(sys.argv[1], r['Name'].split('(')[0][-56:], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
done
