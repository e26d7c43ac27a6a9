#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lpf or bpf or filter or sos or pd_ or edfa or PD or EDFA or chain or transmitter" 2>&1 | tail -4
python tools/sos_prof.py 2>&1 | tail -12 | tee gpurun_out/r2_sos_times.txt
SOS_TWO_KERNELS=1 python tools/sos_prof.py 2>&1 | tail -12 | tee gpurun_out/r2_sos_times_two_kernels.txt
rm -rf gpurun_out/sosp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sosp -- python3 tools/sos_prof.py > /dev/null 2>&1
find gpurun_out/sosp -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-220 | tee gpurun_out/r2_sos_kernel_stats.csv
find gpurun_out/sosp -name "*kernel_trace.csv" -delete
