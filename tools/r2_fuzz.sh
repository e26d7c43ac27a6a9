#!/bin/bash
cd $GRAFT_REPO_ROOT
python tests/diag/fuzz_many.py 400 2026 2>&1 | tail -8 | tee gpurun_out/r2_fuzz_many.txt
python tests/diag/fuzz_filters.py 2>&1 | tail -4 | tee gpurun_out/r2_fuzz_filters.txt
python tests/diag/fuzz_misc.py 2>&1 | tail -6 | tee gpurun_out/r2_fuzz_misc.txt
