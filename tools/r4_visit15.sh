set -u
export TMPDIR=/tmp
bash tools/ab.sh r04v15 3 "c1_phase_f64:PREC=c128 STEPS=100" "c1_complex_table:PREC=c128 STEPS=100 SSFM_PHASE_TABLE=0" "c1_phase_2fields:PREC=c128 STEPS=100 FIELDS=2" "c1_table_2fields:PREC=c128 STEPS=100 FIELDS=2 SSFM_PHASE_TABLE=0" "c128_2e16:PREC=c128 LOG2N=16 STEPS=300" "c128_2e16_table:PREC=c128 LOG2N=16 STEPS=300 SSFM_PHASE_TABLE=0"
python -m pytest tests -m gpu -q -x 2>&1 | tail -6
