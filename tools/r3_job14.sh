#!/bin/bash
# final evidence of the round: parity suite, then profiles + default bench lines of every config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest14.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3_gputest14.log
bash tools/r3_job5.sh "$@"
