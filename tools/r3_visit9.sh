#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3i
: > ${T}_ab.txt
for r in 1 2 3; do
  python tools/step_time.py base >> ${T}_ab.txt 2>&1
  for v in latetab latep lateboth; do SSFM_LIB=build/var/_ssfm_$v.so python tools/step_time.py $v >> ${T}_ab.txt 2>&1; done
  FIELDS=4 python tools/step_time.py base_4fields >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_LIB=build/var/_ssfm_lateboth.so python tools/step_time.py lateboth_4fields >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-60
SSFM_LIB=build/var/_ssfm_trace.so SSFM_GRAPH=0 python tools/trace_timeline.py > ${T}_timeline.txt 2>&1; tail -12 ${T}_timeline.txt
