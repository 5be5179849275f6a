#!/bin/bash
# round 4 job 32: rehearsal of the multi-rank bench lines on one GPU (ranks share the device, gloo carries the reductions): configs 3, 4, 5 at 4 ranks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for C in 3 4 5; do
  R=8; [ $C = 5 ] && R=4
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --config $C --real $R --particles 8000000 --steps 4 --warmup 2 > gpurun_out/r4_j32_c${C}_n4.json 2> gpurun_out/r4_j32_c${C}_n4.err || { echo "FAILED $C"; tail -20 gpurun_out/r4_j32_c${C}_n4.err; exit 1; }
  timeout -k 10 300 python bench.py --config $C --real $R --particles 8000000 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/r4_j32_c${C}_n1.json 2> gpurun_out/r4_j32_c${C}_n1.err || { echo "FAILED n1 $C"; tail -20 gpurun_out/r4_j32_c${C}_n1.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_j32_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]
    print(f.split("j32_")[1], d["n_gpus"], "%.4e"%d["value"], "%.2f ms"%d["ms_per_step"], c.get("reduction_transport"), c.get("live_particles_all_ranks"), c.get("numpart_all_ranks"), c.get("gridunc_sum_all_ranks"), c["counters"]["n_due"] if "counters" in c else None)
PY
