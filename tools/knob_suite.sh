#!/bin/bash
# The whole GPU parity suite under every documented environment knob, then the randomised stress runs (tests/diag).
#   bash tools/knob_suite.sh TAG      -> gpurun_out/TAG_knob_suite.txt, gpurun_out/TAG_fuzz.txt
set -u
TAG=${1:-knobs}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_knob_suite.txt
: > $OUT
for k in "SSFM_LANES=1" "SSFM_E=8" "SSFM_EF=8" "SSFM_SMALL=0" "SSFM_ADAPT_FUSED=0" "SSFM_FUSED_PATIENCE_TICKS=-1" "SSFM_PHASE_TABLE=0" "SSFM_FORCE_FLY=1" \
         "SSFM_LANE_THREADS=0" "SSFM_MEDIUM=0" "SSFM_MEDIUM_ADAPT=0" "SSFM_MEDIUM_SPLIT=0" "SSFM_SOS_ONE_LAUNCH=0" "SSFM_SOS_LONG_CHUNK=0" "SSFM_SOS_NEAR=0" "SSFM_SOS_MEET=1" "SSFM_CHIRP_HALF=0" \
         "SSFM_CHIRP_LOOP=python" "SSFM_CHIRP_SMALL=0" "SSFM_LANE_POOL_OFF=1" "SSFM_SPLIT_LOG2M=21"; do
  echo "== $k" >> $OUT; env $k timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^FAILED|passed|failed" >> $OUT
done
cat $OUT
[ "${KNOBS_ONLY:-0}" = 1 ] && exit 0
F=gpurun_out/${TAG}_fuzz.txt
# (every section in full: round 4's summary was cut to its last lines and hid which filter bin carried its two violations)
{ echo "== tests/diag/fuzz_many.py 400 2026"; timeout 1500 python tests/diag/fuzz_many.py 400 2026 2>&1 | grep -v Warning;
  echo "== tests/diag/fuzz_many.py 400 7"; timeout 1500 python tests/diag/fuzz_many.py 400 7 2>&1 | grep -v Warning;
  echo "== tests/diag/fuzz_filters.py"; timeout 600 python tests/diag/fuzz_filters.py 2>&1;
  echo "== tests/diag/fuzz_misc.py"; timeout 600 python tests/diag/fuzz_misc.py 2>&1; } > $F 2>&1
cat $F
