set -u
export TMPDIR=/tmp
V=build/var
bash tools/ab.sh r04v3 2 "all:" "ablx1:SSFM_LIB=$V/_ssfm_ablx1.so" "ablx2:SSFM_LIB=$V/_ssfm_ablx2.so" "ablnofft:SSFM_LIB=$V/_ssfm_ablnofft.so" "ablmem:SSFM_LIB=$V/_ssfm_ablmem.so" "all_1p:POL=1" "ablx2_1p:SSFM_LIB=$V/_ssfm_ablx2.so POL=1" "ablnofft_1p:SSFM_LIB=$V/_ssfm_ablnofft.so POL=1" "ablmem_1p:SSFM_LIB=$V/_ssfm_ablmem.so POL=1"
