#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/pmc_scatter.sh c5s --config 5 --real 4 --particles 2e7 --global-particles 1e8 --steps 3 --warmup 1
cat gpurun_out/probe_c5s/names.txt | tr '\n' ' ' | cut -c1-3000
python tools/pmc_probe_show.py c5s k_wetdepo
python tools/pmc_probe_show.py c5s k_conccalc
