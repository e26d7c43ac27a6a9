#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3c
: > ${T}_ab.txt
for r in 1 2 3; do
  python tools/step_time.py base >> ${T}_ab.txt 2>&1
  SSFM_LIB=build/var/_ssfm_scalar.so python tools/step_time.py scalar >> ${T}_ab.txt 2>&1
  SSFM_EF=8 SSFM_LIB=build/var/_ssfm_scalar.so python tools/step_time.py scalar_EF8 >> ${T}_ab.txt 2>&1
  POL=1 SSFM_LIB=build/var/_ssfm_scalar.so python tools/step_time.py scalar_1pol >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_LIB=build/var/_ssfm_scalar.so python tools/step_time.py scalar_4fields >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_EF=8 SSFM_LIB=build/var/_ssfm_scalar.so python tools/step_time.py scalar_EF8_4fields >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_EF=8 python tools/step_time.py EF8_4fields >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-64
