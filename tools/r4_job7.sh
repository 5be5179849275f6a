#!/bin/bash
# round 4, GPU job 7: whole suite with the new switches (turboff, interpolhmix, domain fill, quasi-Lagrangian, no kernel), blend determinism
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest7.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4_gputest7.log
