#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_u16b_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_u16b_pytest.log
tools/variants.sh run u16b product nou16 nocap sc1 sc1p twnc sc1p_twnc > /dev/null
cat gpurun_out/var_u16b.txt
