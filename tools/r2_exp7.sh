#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in twnc sc1 sc1p ntl; do echo "== $v"; SSFM_LIB=build/var/_ssfm_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_fft_against_numpy and 0-" 2>&1 | tail -8; done
