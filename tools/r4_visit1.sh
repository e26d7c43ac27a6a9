set -u
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or c2 or C2 or oracle" 2>&1 | tail -5
V=build/var
bash tools/ab.sh r04v1 3 "base:SSFM_LIB=$V/_ssfm_base.so" "pair:SSFM_LIB=$V/_ssfm_pair.so" "micro:SSFM_LIB=$V/_ssfm_micro.so" "p16:SSFM_LIB=$V/_ssfm_p16.so" "all:" "alldb:SSFM_LIB=$V/_ssfm_alldb.so" "base_4f:SSFM_LIB=$V/_ssfm_base.so FIELDS=4" "all_4f:FIELDS=4" "base_1p:SSFM_LIB=$V/_ssfm_base.so POL=1" "all_1p:POL=1"
python -m pytest tests -m gpu -q 2>&1 | tail -8
