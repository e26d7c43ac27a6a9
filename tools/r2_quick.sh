#!/bin/bash
# quick check: FFT + golden parity subset, then two bench lines
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_against_numpy or golden or full_size_c2" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f us/step'%d['us_per_ssfm_step'], {k: round(v,2) for k,v in r['launch_us'].items()}, 'C1 %.1f us' % d['secondary']['us_per_ssfm_step'])"; done
