#!/bin/bash
# round 4 job 30: parity tests after the initialize() wind-pack choice
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_release.py -m gpu -x -q > gpurun_out/r4_gputest30.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_gputest30.log
