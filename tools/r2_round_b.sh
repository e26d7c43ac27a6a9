#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r2b
bash tools/gpu_pmc.sh r2b_pmc
LANES=2 bash tools/gpu_pmc2.sh r2b_sq2 | tee gpurun_out/r2b_sq_lanes2.txt
python tools/cfg_times.py 2>&1 | tee gpurun_out/r2b_cfg_times.txt
python tools/c34_bench.py 2>&1 | tee gpurun_out/r2b_c34.txt
