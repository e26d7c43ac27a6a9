#!/bin/bash
# Round-3 GPU visit 1: parity suite, the bench line, phase-table / fly / ablation A/B on the final build, C3 and C4 bench lines.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=gpurun_out/r3a
python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${T}_pytest.log
python bench.py --steps 5 --warmup 1 > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$?"; cut -c1-3000 ${T}_bench.json; tail -3 ${T}_bench.err
: > ${T}_ab.txt
for r in 1 2 3; do
  SSFM_PHASE_TABLE=1 python tools/step_time.py phase >> ${T}_ab.txt 2>&1
  SSFM_PHASE_TABLE=0 python tools/step_time.py table >> ${T}_ab.txt 2>&1
  SSFM_FORCE_FLY=1 python tools/step_time.py fly >> ${T}_ab.txt 2>&1
  for v in notab nop notw nofft memonly notabnop; do
    SSFM_LIB=build/var/_ssfm_$v.so python tools/step_time.py abl_$v >> ${T}_ab.txt 2>&1
  done
  SSFM_PHASE_TABLE=0 SSFM_LIB=build/var/_ssfm_notab.so python tools/step_time.py abl_notab_complextable >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_PHASE_TABLE=1 python tools/step_time.py phase_4fields >> ${T}_ab.txt 2>&1
  FIELDS=4 SSFM_PHASE_TABLE=0 python tools/step_time.py table_4fields >> ${T}_ab.txt 2>&1
  SSFM_LANES=1 SSFM_PHASE_TABLE=1 python tools/step_time.py phase_1lane >> ${T}_ab.txt 2>&1
  SSFM_LANES=1 SSFM_PHASE_TABLE=0 python tools/step_time.py table_1lane >> ${T}_ab.txt 2>&1
done
sort ${T}_ab.txt | cut -c1-60
python bench.py --workload c3 --steps 2 --warmup 1 --cpu-steps 0 > ${T}_c3.json 2> ${T}_c3.err; echo "c3 rc=$?"; cut -c1-1500 ${T}_c3.json; tail -3 ${T}_c3.err
python bench.py --workload c4 --steps 2 --warmup 1 --cpu-steps 0 > ${T}_c4.json 2> ${T}_c4.err; echo "c4 rc=$?"; cut -c1-1500 ${T}_c4.json; tail -3 ${T}_c4.err
