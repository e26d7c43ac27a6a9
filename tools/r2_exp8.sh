#!/bin/bash
cd $GRAFT_REPO_ROOT
SSFM_LIB=build/var/_ssfm_sc1p_twnc_ntl.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not c3_batch" > gpurun_out/r2_sc1p_twnc_ntl_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_sc1p_twnc_ntl_pytest.log
tools/variants.sh run u16e sc1p_twnc_ntl sc1p_twnc twnc_ntl pwt_twnc_ntl > /dev/null
cat gpurun_out/var_u16e.txt
