#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python tools/big_n_c128.py 2>&1 | grep "2^20 x 2" | sed 's/^/default: /'
SSFM_LANES=1 python tools/big_n_c128.py 2>&1 | grep "2^20 x 2" | sed 's/^/lanes=1: /'
SSFM_LIB=$PWD/build/var/_ssfm_c128c16.so python tools/big_n_c128.py 2>&1 | grep "2^20 x 2" | sed 's/^/cols16: /'
done | tee gpurun_out/r2_c1b.txt
