#!/bin/bash
# Round 6 A/B: the per-thread gate of the 16-bit stale |A|^2 (a sign test and an unlikely branch in the column kernel) -- does the headline notice?
for r in 1 2 3; do
  python3 tools/step_time.py "product   "
  SSFM_LIB=$PWD/build/var/_ssfm_p16gate.so python3 tools/step_time.py "p16 gate  "
done
SSFM_LIB=$PWD/build/var/_ssfm_p16gate.so SSFM_MARGINS_FILE=/tmp/m_gate.txt python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "16_bit_stale or full_size_c2 or thousand_steps or full_size_properties" 2>&1 | tail -1
grep -E "rad per half step|C2 full size" /tmp/m_gate.txt | awk -F'|' '{print "   ", $2, "|", $3, "|", $4, "|", $5}'
