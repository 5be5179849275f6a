#!/bin/bash
# round 4 job 45: the Langevin kernel's grid as the engine sets it (occupancy query)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config 3 --particles 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --opt verbose=1 2>&1 | grep -i "Langevin kernel\|blocks per CU" | head -3
timeout -k 10 300 python bench.py --config 5 --real 4 --particles 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --opt verbose=1 2>&1 | grep -i "Langevin kernel\|blocks per CU" | head -3
