#!/bin/bash
# compare SSFM_LANES settings in one GPU visit (interleaved, 2 rounds)
for r in 1 2; do for L in 1 2; do
  echo -n "lanes=$L: "; SSFM_LANES=$L python bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-profile-pass | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_ssfm_step'], d['value']/1e9)"
done; done
