#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3n
python -m pytest tests -m gpu -x -q -k "sosfilt or filter or LPF or BPF or lpf or bpf" > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${T}_pytest.log
: > ${T}_shapes.txt
for r in 1 2; do for v in product sos_apply2; do
  L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so
  echo "== $v" >> ${T}_shapes.txt
  SSFM_LIB=$L python tools/filter_shapes.py 2>&1 >> ${T}_shapes.txt
done; done
cat ${T}_shapes.txt
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 tools/sos_prof.py > ${T}_sosprof.txt 2> ${T}_prof.err; cat ${T}_sosprof.txt
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_sos_kernel_stats.csv; grep "k_chunk\|k_apply" ${T}_sos_kernel_stats.csv | cut -c1-60,200-320
find ${T}_prof -name "*kernel_trace.csv" -size +1M -delete
