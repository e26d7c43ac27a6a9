#!/bin/bash
# points per thread of the two kernels under the round-2 memory policy (C2, bench.py)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" "SSFM_E=8" "SSFM_EF=8" "SSFM_E=8 SSFM_EF=8"; do
  echo -n "[$v] "
  env $v python bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % d['us_per_ssfm_step'])"
done; done 2>&1 | tee gpurun_out/r2_e8.txt
