#!/bin/bash
# The adaptive run at 2^20 x 2 (the reference's default mode at the headline size): kernel stats, SQ counters and PMC traffic of its two kernels
# (k_time<TM_MID_A> over 512 workgroups, k_freq<FM_FLY> over both rows), and the stagger experiment.   bash tools/r4_adaptive_round.sh TAG
set -u
TAG=${1:-r04_adaptive}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/$TAG
V=build/var
PROBE="python tools/adaptive_prof.py" bash tools/ab.sh ${TAG} 3 "product:" "three_launches:SSFM_ADAPT_FUSED=0"
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 tools/adaptive_prof.py > ${T}_run.txt 2> ${T}_prof.err
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_kernel_stats.csv; head -6 ${T}_kernel_stats.csv | cut -c1-220
find ${T}_prof -name "*kernel_trace.csv" -size +1M -delete
rm -rf ${T}_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d ${T}_sq -- python3 tools/adaptive_prof.py > /dev/null 2> ${T}_sq.err
python tools/sq_summary.py ${T}_sq "k_time_mid_a_c64=k_time<float, 256, 16, 16, 6" "k_freq_fly_c64=k_freq<float, 4096, 1, 16, 1" > ${T}_sq.txt; cat ${T}_sq.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf ${T}_$C
  rocprofv3 --pmc $C --output-format csv -d ${T}_$C -- python3 tools/adaptive_prof.py > /dev/null 2> ${T}_$C.err
  python tools/sq_summary.py ${T}_$C "k_time_mid_a_c64=k_time<float, 256, 16, 16, 6" "k_freq_fly_c64=k_freq<float, 4096, 1, 16, 1" >> ${T}_sq.txt
done
tail -4 ${T}_sq.txt
find ${T}_sq ${T}_FETCH_SIZE ${T}_WRITE_SIZE -name "*.csv" -size +1M -delete
