#!/bin/bash
cd $GRAFT_REPO_ROOT
SSFM_LIB=build/var/_ssfm_ldsswz.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_against_numpy or golden or full_size_c2 or adaptive" 2>&1 | tail -2
FOUR=0 tools/variants.sh run lds nolds ldsswz > /dev/null; cat gpurun_out/var_lds.txt
SSFM_LIB=build/var/_ssfm_ldsswz.so LANES=2 bash tools/gpu_pmc2.sh r2_sq_swz 2>&1 | tail -2
