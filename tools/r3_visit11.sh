#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3k
for k in "SSFM_LANES=1" "SSFM_FUSED_PATIENCE_TICKS=-1" "SSFM_FORCE_FLY=1"; do
  echo "== $k"; env $k timeout 900 python -m pytest tests -m gpu -q -rf 2>&1 | tail -12
done > ${T}_knobfail.txt 2>&1
cat ${T}_knobfail.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for r in 1 2 3; do python tools/step_time.py base_latep; done
