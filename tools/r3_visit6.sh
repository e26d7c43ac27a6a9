#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3f
: > ${T}_shapes.txt
for r in 1 2; do for v in product sos_c8 sos_c16 sos_c24 sos_w2 sos_c16w2 sos_c24w2; do
  L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so
  echo "== $v" >> ${T}_shapes.txt
  SSFM_LIB=$L python tools/filter_shapes.py 2>&1 | head -2 >> ${T}_shapes.txt
done; done
cat ${T}_shapes.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"; do
  rm -rf ${T}_sq
  rocprofv3 --pmc $set --output-format csv -d ${T}_sq -- python3 tools/sos_prof.py > /dev/null 2> ${T}_sq.err
  python tools/sq_summary.py ${T}_sq "chunk_scan_c=k_chunk_scan<2, 2, 4>" "apply_c=k_apply<2, 2, 4>" >> ${T}_sq.txt
done
cat ${T}_sq.txt
find ${T}_sq -name "*.csv" -size +1M -delete
