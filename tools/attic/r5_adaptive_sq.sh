#!/bin/bash
# round 5: SQ counters of the adaptive run's two kernels at 2^20 x 2 (k_time<TM_MID_A> over 512 workgroups, k_freq<FM_FLY_IM>), after the poll-loop change
export TMPDIR=/tmp; cd "$(dirname "$0")/../.."; T=gpurun_out/r05_adaptive
rm -rf ${T}_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d ${T}_sq -- python3 tools/adaptive_prof.py > /dev/null 2> ${T}_sq.err
python tools/sq_summary.py ${T}_sq "k_time_mid_a_c64=k_time<float, 256, 16, 16, 6" "k_freq_fly_c64=k_freq<float, 4096, 1, 16, 4" > ${T}_sq.txt
find ${T}_sq -name "*.csv" -size +1M -delete
cat ${T}_sq.txt
