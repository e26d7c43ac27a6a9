#!/bin/bash
# the whole GPU suite under the documented environment knobs
cd $GRAFT_REPO_ROOT
for k in "SSFM_LANES=1" "SSFM_GRAPH=1" "SSFM_E=8" "SSFM_GRAPH=auto" "SSFM_STAGGER=1"; do
  echo "== $k"; env $k python -m pytest tests -m gpu -q -x 2>&1 | tail -2
done 2>&1 | tee gpurun_out/r2_knob_suite.txt
