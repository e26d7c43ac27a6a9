#!/bin/bash
# Filter memory policy A/B: write-through stores (SOS_WT) and non-temporal loads (SOS_NT) in sos_filter.hip.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r2_sos_policy.txt; : > $OUT
for r in 1 2; do for v in product soswt sosnt soswtnt; do
  L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so
  echo "== $v" >> $OUT
  SSFM_LIB=$PWD/$L python tools/filter_shapes.py >> $OUT 2>&1
done; done
SSFM_LIB=$PWD/build/var/_ssfm_soswtnt.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lpf or bpf or filter or sos" 2>&1 | tail -3 >> $OUT
cat $OUT
