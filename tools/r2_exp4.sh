#!/bin/bash
cd $GRAFT_REPO_ROOT
for mode in 0 1 2 4 6 5; do timeout 120 build/ub/persist_probe 300 0 256 $mode; done 2>&1 | tee gpurun_out/r2_persist_probe2.txt
