set -u
export TMPDIR=/tmp
V=build/var
bash tools/gpu_round.sh r04a 2>&1 | tail -60
bash tools/ab.sh r04_c1 2 "c1:PREC=c128 STEPS=100" "c1_policy:PREC=c128 STEPS=100 SSFM_LIB=$V/_ssfm_c128pol.so" "c1_lanes1:PREC=c128 STEPS=100 SSFM_LANES=1" "c1_e16:PREC=c128 STEPS=100 SSFM_E=16" "c1_ef8:PREC=c128 STEPS=100 SSFM_EF=8" "c1_cols16:PREC=c128 STEPS=100 SSFM_LIB=$V/_ssfm_c128cols16.so" "c1_2fields:PREC=c128 STEPS=100 FIELDS=2"
