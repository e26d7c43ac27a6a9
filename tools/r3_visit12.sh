#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3l
timeout 600 python -m pytest tests -m gpu -x -q -k "chained" > ${T}_pytest_chain.log 2>&1; echo "pytest chained rc=$?"; tail -15 ${T}_pytest_chain.log
: > ${T}_ab.txt
for r in 1 2 3; do
  SSFM_CHAIN=1 timeout 120 python tools/step_time.py chain >> ${T}_ab.txt 2>&1
  SSFM_CHAIN=0 timeout 120 python tools/step_time.py plain >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_CHAIN=1 timeout 120 python tools/step_time.py chain_4fields >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_CHAIN=0 timeout 120 python tools/step_time.py plain_4fields >> ${T}_ab.txt 2>&1
  FIELDS=2 SSFM_CHAIN=1 timeout 120 python tools/step_time.py chain_2fields >> ${T}_ab.txt 2>&1
  FIELDS=2 SSFM_CHAIN=0 timeout 120 python tools/step_time.py plain_2fields >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-64
