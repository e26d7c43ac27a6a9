#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adaptive or golden or return_steps or snapshot or progress or chain or random_parameters or rccl" 2>&1 | tail -3
for r in 1 2; do echo -n "two lanes: "; python tools/adaptive_prof.py 2>&1 | tail -1; echo -n "one lane:  "; SSFM_ADAPT_ONE_LANE=1 python tools/adaptive_prof.py 2>&1 | tail -1; done
python tools/small_n.py 2>&1 | tail -4
