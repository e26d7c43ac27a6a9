#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3m
timeout 900 python -m pytest tests -m gpu -x -q -k "medium" > ${T}_pytest_medium.log 2>&1; echo "pytest medium rc=$?"; tail -15 ${T}_pytest_medium.log
: > ${T}_ab.txt
for r in 1 2; do for k in 14 15 16 17; do for pol in 1 2; do
  LOG2N=$k POL=$pol SSFM_MEDIUM=1 timeout 120 python tools/step_time.py medium_2^${k}x${pol} >> ${T}_ab.txt 2>&1
  LOG2N=$k POL=$pol SSFM_MEDIUM=0 timeout 120 python tools/step_time.py twokernel_2^${k}x${pol} >> ${T}_ab.txt 2>&1
done; done; done
sort ${T}_ab.txt | cut -c1-70
timeout 900 python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${T}_pytest.log
