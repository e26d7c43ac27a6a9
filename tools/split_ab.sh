#!/bin/bash
# Round 6 A/B of the split plans' row pass: the product, + the unit layout between the two halves (16-byte accesses), + 4-byte operator phases.   bash tools/split_ab.sh
set -u
for r in 1 2 3; do
  echo "== round $r"
  for name in product splitu16 splitu16ph; do
    L=$PWD/build/var/_ssfm_$name.so; [ $name = product ] && L=$PWD/opticomlib_amd/_ssfm_amd.so
    for k in 23 24; do echo -n "$name: "; SSFM_LIB=$L python3 tools/big_n_run.py $k 100; done
  done
done
echo "== parity of the variants (the split-plan tests of the suite)"
for name in splitu16 splitu16ph; do
  SSFM_LIB=$PWD/build/var/_ssfm_$name.so SSFM_MARGINS_FILE=/tmp/m_$name.txt python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "split_plans or two_to_the_23 or two_to_the_24" 2>&1 | tail -1
  grep -E "oracle fixture|oracle, 2\^21" /tmp/m_$name.txt | awk -F'|' '{print "   ", $2, "|", $4, "|", $5}'
done
