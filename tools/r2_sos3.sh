#!/bin/bash
# Filter workgroup shape under LDS: 4 wavefronts x 12 samples = 54 KB per workgroup (two per CU)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2_sos_shape.txt; : > $OUT
for r in 1 2; do for v in product w2 w3 c10 c10w2; do
  L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so
  echo "== $v" >> $OUT
  SSFM_LIB=$PWD/$L python tools/filter_shapes.py >> $OUT 2>&1
done; done
for v in w2 w3 c10; do echo "== parity $v" >> $OUT; SSFM_LIB=$PWD/build/var/_ssfm_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lpf or bpf or filter or sos" 2>&1 | tail -2 >> $OUT; done
cat $OUT
