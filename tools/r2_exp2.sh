#!/bin/bash
# round 2, experiment 2: U16 layout (16-byte units + permlane32 swap): parity, then A/B against plain / write-through
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_u16_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_u16_pytest.log
SSFM_LIB=build/var/_ssfm_u16twnc.so python bench.py --steps 2 --warmup 1 --cpu-steps 0 2>&1 | tail -5 | cut -c1-600
tools/variants.sh run u16 product nou16 u16sc1 u16sc1p u16twnc > /dev/null
cat gpurun_out/var_u16.txt
