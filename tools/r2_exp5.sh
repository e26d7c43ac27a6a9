#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in tr_base tr_sc1p_twnc tr_nou16; do echo "=== $v"; SSFM_LIB=build/var/_ssfm_$v.so SSFM_GRAPH=0 python tools/trace_timeline.py; echo "=== $v lanes=1"; SSFM_LANES=1 SSFM_LIB=build/var/_ssfm_$v.so SSFM_GRAPH=0 python tools/trace_timeline.py | tail -6; done > gpurun_out/r2_timeline.txt 2>&1
cat gpurun_out/r2_timeline.txt
