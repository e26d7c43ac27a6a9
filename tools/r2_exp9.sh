#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in sc1; do echo "== $v"; SSFM_LIB=build/var/_ssfm_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_fft_against_numpy and 0-" 2>&1 | tail -8; done
SSFM_LIB=build/var/_ssfm_sc1p_twnc_ntl.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not c3_batch" > gpurun_out/r2_sc1p_twnc_ntl_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2_sc1p_twnc_ntl_pytest.log
