#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in product stag40 stag80 stag120; do L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so
echo -n "$v lanes=1 fixed: "; SSFM_LANES=1 SSFM_LIB=$L python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step'%d['us_per_ssfm_step'])"
echo -n "$v adaptive: "; SSFM_LIB=$L python tools/adaptive_prof.py 2>&1 | tail -1
done; done
