#!/bin/bash
# Interleaved A/B rounds of tools/step_time.py (or another probe) on the GPU box.
#   bash tools/ab.sh TAG ROUNDS "label:ENV=val ENV=val" "label2:SSFM_LIB=build/var/_ssfm_x.so FIELDS=4" ...
#   PROBE="python tools/adaptive_prof.py" bash tools/ab.sh ...      (default probe: python tools/step_time.py LABEL)
# -> gpurun_out/TAG_ab.txt (sorted copy printed).  A label's settings are plain environment assignments; `product` = none.
set -u
TAG=$1; ROUNDS=$2; shift 2
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab.txt
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    label=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
    if [ -n "${PROBE:-}" ]; then echo -n "$label: " >> $OUT; env $envs timeout 300 $PROBE 2>&1 | tail -${PROBE_LINES:-1} >> $OUT
    else env $envs timeout 300 python tools/step_time.py $label >> $OUT 2>&1; fi
  done
done
sort $OUT | cut -c1-110
