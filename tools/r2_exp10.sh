#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r2_full_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2_full_pytest.log
python bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | cut -c1-700
