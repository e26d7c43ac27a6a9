#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3j
./build/ubench_exchange > ${T}_ubench_exchange.txt 2>&1; cat ${T}_ubench_exchange.txt
: > ${T}_ab.txt
for r in 1 2 3 4 5 6; do
  REPS=5 python tools/step_time.py base >> ${T}_ab.txt 2>&1
  for v in latep lateboth; do REPS=5 SSFM_LIB=build/var/_ssfm_$v.so python tools/step_time.py $v >> ${T}_ab.txt 2>&1; done
done
sort ${T}_ab.txt | cut -c1-60
bash tools/knob_suite.sh r3j
