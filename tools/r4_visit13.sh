set -u
export TMPDIR=/tmp
V=build/var
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/ab.sh r04v13 3 "preload:" "nopreload:SSFM_LIB=$V/_ssfm_nopreload.so" "preload_1p:POL=1" "nopreload_1p:SSFM_LIB=$V/_ssfm_nopreload.so POL=1" "preload_2e16:LOG2N=16 POL=1" "nopreload_2e16:SSFM_LIB=$V/_ssfm_nopreload.so LOG2N=16 POL=1" "preload_c128:PREC=c128 STEPS=100" "nopreload_c128:SSFM_LIB=$V/_ssfm_nopreload.so PREC=c128 STEPS=100"
PROBE="python tools/adaptive_prof.py" bash tools/ab.sh r04v13a 2 "preload:" "nopreload:SSFM_LIB=$V/_ssfm_nopreload.so"
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
