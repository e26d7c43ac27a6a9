#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in sm8 sm16; do echo "== $v"; SSFM_LIB=$PWD/build/var/_ssfm_$v.so python tools/small_engine.py 2>&1 | grep "^c64"; done | tee gpurun_out/r2_small_engine_e.txt
