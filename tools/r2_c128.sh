#!/bin/bash
cd $GRAFT_REPO_ROOT
SSFM_LIB=build/var/_ssfm_c128twc.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_against_numpy or c128 or complex128 or twin or dm_ or DM or c1_" 2>&1 | tail -2
for i in 1 2; do for v in product c128twc c128wt; do L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so; echo -n "$v: "; SSFM_LIB=$L python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 %.2f us/step'%d['us_per_ssfm_step'], 'C1 %.1f us' % d['secondary']['us_per_ssfm_step'])"; done; done
