#!/bin/bash
# Round-3 GPU visit 2: issue-rate microbenchmark, knob A/B on the phase-table build, kernel stats, PMC traffic, SQ counters.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3b
./build/ubench_pk > ${T}_ubench_pk.txt 2>&1; cat ${T}_ubench_pk.txt
: > ${T}_ab.txt
for r in 1 2 3; do
  python tools/step_time.py base >> ${T}_ab.txt 2>&1
  SSFM_E=8 python tools/step_time.py E8_both >> ${T}_ab.txt 2>&1
  SSFM_EF=8 python tools/step_time.py EF8_freq_only >> ${T}_ab.txt 2>&1
  SSFM_STAGGER=1 python tools/step_time.py stagger >> ${T}_ab.txt 2>&1
  SSFM_GRAPH=1 python tools/step_time.py graph >> ${T}_ab.txt 2>&1
  SSFM_LIB=build/var/_ssfm_w3.so python tools/step_time.py w3 >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_LIB=build/var/_ssfm_w3.so python tools/step_time.py w3_4fields >> ${T}_ab.txt 2>&1
  FIELDS=2 python tools/step_time.py base_2fields >> ${T}_ab.txt 2>&1
  FIELDS=2 SSFM_LANES=4 python tools/step_time.py base_2fields_4lanes >> ${T}_ab.txt 2>&1
  POL=1 python tools/step_time.py base_1pol >> ${T}_ab.txt 2>&1
  PREC=c128 STEPS=100 python tools/step_time.py c128 >> ${T}_ab.txt 2>&1
  PREC=c128 STEPS=100 SSFM_EF=8 python tools/step_time.py c128_EF8 >> ${T}_ab.txt 2>&1
  PREC=c128 STEPS=100 SSFM_E=16 python tools/step_time.py c128_E16 >> ${T}_ab.txt 2>&1
  PREC=c128 STEPS=100 SSFM_LANES=1 python tools/step_time.py c128_1lane >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-64
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > ${T}_prof_bench.json 2> ${T}_prof.err; echo "rocprof rc=$?"
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_kernel_stats.csv; head -12 ${T}_kernel_stats.csv
find ${T}_prof -name "*kernel_trace.csv" -size +2M -delete
bash tools/gpu_pmc.sh r3b_pmc > ${T}_pmc.log 2>&1; tail -30 ${T}_pmc.log
LANES=2 bash tools/gpu_pmc2.sh r3b_sq > ${T}_sq_lanes2.txt 2>&1; cat ${T}_sq_lanes2.txt
LANES=2 CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" bash tools/gpu_pmc2.sh r3b_sq2 > ${T}_sq2_lanes2.txt 2>&1; cat ${T}_sq2_lanes2.txt
