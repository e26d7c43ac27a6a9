set -u
export TMPDIR=/tmp
V=build/var
bash tools/ab.sh r04v6 3 "ka:" "noka:SSFM_LIB=$V/_ssfm_noka.so" "ka_optflush0:AMD_OPT_FLUSH=0" "ka_hwq8:GPU_MAX_HW_QUEUES=8" "ka_hwq2:GPU_MAX_HW_QUEUES=2" "ka_nosdma:HSA_ENABLE_SDMA=0" "ka_1p:POL=1" "noka_1p:SSFM_LIB=$V/_ssfm_noka.so POL=1"
