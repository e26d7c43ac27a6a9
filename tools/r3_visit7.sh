#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3g
timeout 600 python -m pytest tests -m gpu -x -q -k "adaptive or two_lane" > ${T}_pytest_adapt.log 2>&1; echo "pytest adaptive rc=$?"; tail -5 ${T}_pytest_adapt.log
: > ${T}_ab.txt
for r in 1 2 3; do
  echo "lanes2:" >> ${T}_ab.txt; timeout 120 python tools/adaptive_prof.py >> ${T}_ab.txt 2>&1
  echo "lanes1:" >> ${T}_ab.txt; SSFM_ADAPT_LANES=1 timeout 120 python tools/adaptive_prof.py >> ${T}_ab.txt 2>&1
done
cat ${T}_ab.txt
rm -rf ${T}_adprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_adprof -- python3 tools/adaptive_prof.py > /dev/null 2> ${T}_adprof.err
find ${T}_adprof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_adaptive_kernel_stats.csv; head -6 ${T}_adaptive_kernel_stats.csv
find ${T}_adprof -name "*kernel_trace.csv" -size +1M -delete
timeout 900 python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${T}_pytest.log
