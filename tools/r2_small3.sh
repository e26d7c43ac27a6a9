#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/small_adaptive.py 2>&1 | tee gpurun_out/r2_small_adaptive.txt
