#!/bin/bash
# The single-launch engine of medium plans (k_medium) against the two-kernel engine (GPU box): parity tests, us per step, the whole GPU suite.
#   bash tools/medium_round.sh TAG   -> gpurun_out/TAG_medium_*
set -u
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
T=gpurun_out/${TAG}_medium
timeout 900 python -m pytest tests -m gpu -x -q -k "medium" > ${T}_pytest_medium.log 2>&1; echo "pytest medium rc=$?"; tail -5 ${T}_pytest_medium.log
: > ${T}_ab.txt
for r in 1 2; do for k in 14 15 16 17; do for pol in 1 2; do
  LOG2N=$k POL=$pol SSFM_MEDIUM=1 timeout 120 python tools/step_time.py medium_2^${k}x${pol} >> ${T}_ab.txt 2>&1
  LOG2N=$k POL=$pol SSFM_MEDIUM=0 timeout 120 python tools/step_time.py twokernel_2^${k}x${pol} >> ${T}_ab.txt 2>&1
done; done; done
sort ${T}_ab.txt | cut -c1-70
timeout 900 python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${T}_pytest.log
