set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -25
python tools/step_time.py prod
V=build/var
PROBE="python tools/adaptive_prof.py" bash tools/ab.sh r04v10 2 "adapt:" "noadapt:SSFM_LIB=$V/_ssfm_noadapt.so" "nosplit:SSFM_LIB=$V/_ssfm_nosplit.so" "adapt3:SSFM_ADAPT_FUSED=0"
