#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adaptive or golden or return_steps or snapshot or progress or chain or random_parameters" 2>&1 | tail -5
python tools/cfg_times.py 2>&1 | head -3
python tools/small_n.py 2>&1 | tail -6
rm -rf gpurun_out/adp2; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/adp2 -- python3 tools/adaptive_prof.py > gpurun_out/r2_adaptive_prof.txt 2>&1
find gpurun_out/adp2 -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200 | tee -a gpurun_out/r2_adaptive_prof.txt
find gpurun_out/adp2 -name "*kernel_trace.csv" -delete
