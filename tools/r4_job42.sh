#!/bin/bash
# round 4 job 42: full GPU suite + smoke() + the default bench line on the final code
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest42.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gputest42.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout -k 10 900 python bench.py ) > gpurun_out/r4_j42_default.json 2> gpurun_out/r4_j42_default.err; echo "bench rc=$?"; tail -4 gpurun_out/r4_j42_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_j42_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","scaling","dtype","vs_baseline")}); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu"]["frac_of_launch"], d["cpu_baseline"])
PY
