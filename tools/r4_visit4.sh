set -u
export TMPDIR=/tmp
V=build/var
bash tools/ab.sh r04v4 3 "head:" "nohead:SSFM_LIB=$V/_ssfm_nohead.so" "head_1p:POL=1" "nohead_1p:SSFM_LIB=$V/_ssfm_nohead.so POL=1" "head_4f:FIELDS=4" "head_c128:PREC=c128 STEPS=100" "nohead_c128:SSFM_LIB=$V/_ssfm_nohead.so PREC=c128 STEPS=100" "head_2e16:LOG2N=16 POL=1" "nohead_2e16:SSFM_LIB=$V/_ssfm_nohead.so LOG2N=16 POL=1" "head_2e18:LOG2N=18" "nohead_2e18:SSFM_LIB=$V/_ssfm_nohead.so LOG2N=18"
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
