#!/bin/bash
# Evidence for the zero-phase filter (GPU box): kernel stats, SQ counters, the one-launch kernel's phase timeline, both forms side by side.
#   bash tools/sos_round.sh TAG     -> gpurun_out/TAG_sosfilt_*
# (the timeline needs build/var/_ssfm_tl.so: VSRC=sos_filter.hip tools/variants.sh build tl:-DSOS_TIMELINE=1)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${TAG}_sosfilt
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{ for r in 1 2; do
    echo "== one launch where the call fits (default)"; python3 $R/tools/filter_shapes.py
    echo "== three launches (SSFM_SOS_ONE_LAUNCH=0)"; SSFM_SOS_ONE_LAUNCH=0 python3 $R/tools/filter_shapes.py
  done; } > ${T}_forms.txt 2>&1
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 $R/tools/sos_prof.py > ${T}_run.txt 2> ${T}_prof.err
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_kernel_stats.csv
find ${T}_prof -name "*kernel_trace.csv" -size +1M -delete
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf ${T}_sq
  rocprofv3 --pmc $set --output-format csv -d ${T}_sq -- python3 $R/tools/sos_prof.py > /dev/null 2> ${T}_sq.err
  python3 $R/tools/sq_summary.py ${T}_sq "filtfilt_c=k_filtfilt<2, 2, 4" "filtfilt_r=k_filtfilt<2, 1, 1" >> ${T}_sq.txt
done
find ${T}_sq -name "*.csv" -size +1M -delete
if [ -f $R/build/var/_ssfm_tl.so ]; then
  { echo "== 2^20 x 2 complex128"; SSFM_LIB=$R/build/var/_ssfm_tl.so python3 $R/tools/sos_timeline.py
    echo "== 2^16 real"; SSFM_LIB=$R/build/var/_ssfm_tl.so LOG2N=16 ROWS=1 CPLX=0 python3 $R/tools/sos_timeline.py; } > ${T}_timeline.txt 2>&1
fi
python3 $R/tests/diag/fuzz_filters.py > ${T}_fuzz.txt 2>&1
tail -12 ${T}_forms.txt; head -6 ${T}_kernel_stats.csv | cut -c1-180; cat ${T}_sq.txt | cut -c1-400; tail -4 ${T}_fuzz.txt
