#!/bin/bash
# final sanity of the round: whole GPU suite, the default bench line, the 2-rank rehearsal of bench.py on one GPU
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest15.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3_gputest15.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r3_bench15_default.json 2> gpurun_out/r3_bench15_default.err; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --gpus 2 --particles 2e7 --no-cpu-baseline > gpurun_out/r3_bench15_n2.json 2> gpurun_out/r3_bench15_n2.err; echo "bench n2 rc=$?"
python - <<'PY'
import json
for f in ("r3_bench15_default", "r3_bench15_n2"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["config"].get("rccl_nranks"), d["config"].get("reduction_transport"), d["config"].get("live_particles_all_ranks"))
PY
