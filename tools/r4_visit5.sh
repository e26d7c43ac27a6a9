set -u
export TMPDIR=/tmp
V=build/var
bash tools/ab.sh r04v5 3 "early:" "noearly:SSFM_LIB=$V/_ssfm_noearly.so" "latep0:SSFM_LIB=$V/_ssfm_latep0.so" "early_devkernarg1:HIP_FORCE_DEV_KERNARG=1" "early_devkernarg0:HIP_FORCE_DEV_KERNARG=0" "early_1p:POL=1" "noearly_1p:SSFM_LIB=$V/_ssfm_noearly.so POL=1" "early_4f:FIELDS=4"
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
