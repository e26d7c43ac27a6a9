set -u
export TMPDIR=/tmp
bash tools/ab.sh r04v7 2 "prod:" "ef8:SSFM_EF=8" "e8:SSFM_E=8" "prod_4f:FIELDS=4" "ef8_4f:SSFM_EF=8 FIELDS=4"
python bench.py --steps 10 --warmup 2 2>gpurun_out/r04v7_bench.err | tee gpurun_out/r04v7_bench.json | cut -c1-1500
