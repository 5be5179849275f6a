#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "occupancy or aerosol or three_species or multi_release or time_slices or nest" > gpurun_out/r4_gputest25.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4_gputest25.log
