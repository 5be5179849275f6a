#!/bin/bash
# round 4 job 44: particles born after the upload (birth horizon); full GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest44.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4_gputest44.log
