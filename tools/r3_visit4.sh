#!/bin/bash
# Round-3 GPU visit 4: parity after the float64 small-angle sincos and the packed on-the-fly operator; C1 and adaptive timings + counters.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3d
python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${T}_pytest.log
: > ${T}_ab.txt
for r in 1 2 3; do
  PREC=c128 STEPS=100 python tools/step_time.py c128 >> ${T}_ab.txt 2>&1
  SSFM_FORCE_FLY=1 python tools/step_time.py c64_fly_fixed >> ${T}_ab.txt 2>&1
  python tools/step_time.py c64_base >> ${T}_ab.txt 2>&1
  python tools/adaptive_prof.py >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-70
rm -rf ${T}_c1prof ${T}_adprof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_c1prof -- python3 tools/c1_prof.py > /dev/null 2> ${T}_c1prof.err
find ${T}_c1prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_c1_kernel_stats.csv; head -6 ${T}_c1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_adprof -- python3 tools/adaptive_prof.py > /dev/null 2> ${T}_adprof.err
find ${T}_adprof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_adaptive_kernel_stats.csv; head -6 ${T}_adaptive_kernel_stats.csv
find ${T}_c1prof ${T}_adprof -name "*kernel_trace.csv" -size +1M -delete
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf ${T}_c1sq
  rocprofv3 --pmc $set --output-format csv -d ${T}_c1sq -- python3 tools/c1_prof.py > /dev/null 2> ${T}_c1sq.err
  python tools/sq_summary.py ${T}_c1sq "k_freq_c128=k_freq<double" "k_time_mid_c128=k_time<double, 256, 8, 8, 1" >> ${T}_c1_sq.txt
done
cat ${T}_c1_sq.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf ${T}_c1pmc_$C
  rocprofv3 --pmc $C --output-format csv -d ${T}_c1pmc_$C -- python3 tools/c1_prof.py > /dev/null 2> ${T}_c1pmc.err
  python tools/sq_summary.py ${T}_c1pmc_$C "k_freq_c128=k_freq<double" "k_time_mid_c128=k_time<double, 256, 8, 8, 1" >> ${T}_c1_traffic.txt
done
cat ${T}_c1_traffic.txt
find ${T}_c1sq ${T}_c1pmc_FETCH_SIZE ${T}_c1pmc_WRITE_SIZE -name "*.csv" -size +1M -delete
