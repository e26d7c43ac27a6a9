#!/bin/bash
# Round 6, VERDICT r05 item 3: configuration C1 (2^20 x 2 complex128, 100 x 1 km) -- the A/Bs.   bash tools/c1_ab.sh > gpurun_out/r06_c1_ab.txt
# (a) the stale |A|^2 as 32-bit fixed point relative to the thread's maximum (-DSSFM_P32=1): step time, and the C1 fixture's distance (bound 1e-10)
# (c) the second lane started late (-DSSFM_AB_STAGGER=1, SSFM_LANE_STAGGER_US): does a half-period offset between k_time of one polarisation and k_freq of
#     the other survive, and does it pay?
set -u
P=build/var/_ssfm_p32.so; S=build/var/_ssfm_stagger.so
for r in 1 2 3; do
  echo "== round $r"
  PREC=c128 STEPS=100 REPS=5 python3 tools/step_time.py "product                      "
  SSFM_LIB=$PWD/$P PREC=c128 STEPS=100 REPS=5 python3 tools/step_time.py "32-bit stale |A|^2 (P32)     "
  for us in 0 5 10 20 40; do
    SSFM_LIB=$PWD/$S SSFM_LANE_STAGGER_US=$us PREC=c128 STEPS=100 REPS=5 python3 tools/step_time.py "lane 1 started $us us late    "
  done
  PREC=c128 STEPS=1000 REPS=3 python3 tools/step_time.py "product, 1000 steps          "
  SSFM_LIB=$PWD/$S SSFM_LANE_STAGGER_US=20 PREC=c128 STEPS=1000 REPS=3 python3 tools/step_time.py "lane 1 20 us late, 1000 steps"
done
echo "== accuracy of the P32 variant: the C1 full-size fixture (bound 1e-10) and the complex128 suite"
rm -f /tmp/p32_margins.txt
SSFM_LIB=$PWD/$P SSFM_MARGINS_FILE=/tmp/p32_margins.txt python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "c1_against or c128_against or full_size_c128" 2>&1 | tail -3
grep -E "C1 full size|float64 restatement" /tmp/p32_margins.txt | awk -F'|' '{print "   ", $1, "|", $2, "|", $4, "|", $5}' | head -12
echo "== the same with the product"
rm -f /tmp/p32_margins.txt
SSFM_MARGINS_FILE=/tmp/p32_margins.txt python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "c1_against" 2>&1 | tail -1
grep -E "C1 full size" /tmp/p32_margins.txt | awk -F'|' '{print "   ", $1, "|", $2, "|", $4, "|", $5}'
