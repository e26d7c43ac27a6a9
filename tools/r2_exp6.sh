#!/bin/bash
cd $GRAFT_REPO_ROOT
SSFM_LIB=build/var/_ssfm_sc1p_twnc.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not c3_batch" > gpurun_out/r2_sc1p_twnc_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_sc1p_twnc_pytest.log
FOUR=0 tools/variants.sh run u16c sc1p_twnc pwt pwt_twnc sc1p_twnc_ntl > /dev/null
cat gpurun_out/var_u16c.txt
echo "stagger:"; SSFM_STAGGER=1 FOUR=0 tools/variants.sh run u16d sc1p_twnc > /dev/null; cat gpurun_out/var_u16d.txt
