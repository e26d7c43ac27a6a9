#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r2a
bash tools/gpu_pmc.sh r2a_pmc
LANES=2 bash tools/gpu_pmc2.sh r2a_sq2 | tee gpurun_out/r2a_sq_lanes2.txt
LANES=1 bash tools/gpu_pmc2.sh r2a_sq1 | tee gpurun_out/r2a_sq_lanes1.txt
python tools/batch_sweep.py 2>&1 | tee gpurun_out/r2a_batch_sweep.txt
python tools/c34_bench.py 2>&1 | tee gpurun_out/r2a_c34.txt
python tools/cfg_times.py 2>&1 | tee gpurun_out/r2a_cfg_times.txt
