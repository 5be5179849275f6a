#!/bin/bash
# round 4: the whole GPU suite, smoke, the 2-rank rehearsal of bench.py
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest23.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_gputest23.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
