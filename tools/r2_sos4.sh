#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lpf or bpf or filter or sos or pd_ or PD or EDFA or chain or transmitter or mzm" 2>&1 | tail -3
OUT=gpurun_out/r2_sos_shape2.txt; : > $OUT
for r in 1 2; do for v in auto 4 2; do
  echo "== SOS_WAVES_FORCE=$v" >> $OUT
  if [ $v = auto ]; then python tools/filter_shapes.py >> $OUT 2>&1; else SOS_WAVES_FORCE=$v python tools/filter_shapes.py >> $OUT 2>&1; fi
done; done
cat $OUT
