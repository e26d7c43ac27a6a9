#!/bin/bash
# Build named variants of the whole library with extra -D flags (CPU box), or time them (GPU box).
#   tools/variants.sh build  name:-DFLAG=1@-DOTHER=2 ...      -> build/var/_ssfm_<name>.so
#   tools/variants.sh run TAG name ...                        -> gpurun_out/var_TAG.txt
# run: interleaved, two rounds; single field (bench.py, C2) and 4 fields resident per variant.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/var
SRC="ssfm_host.hip sos_filter.hip frontend.hip device_mem.hip chirpz.hip transmitter.hip prbs.hip"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $OUT
  n=0
  for v in "$@"; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr '@' ' ')
    ( cd $ROOT/opticomlib_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$ROOT/include $flags -o $OUT/_ssfm_$name.so $SRC 2>$OUT/$name.log && echo built $name || { echo FAILED $name; tail -5 $OUT/$name.log; } ) &
    n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
  done; wait
else
  TAG=$1; shift
  mkdir -p $ROOT/gpurun_out
  LOG=$ROOT/gpurun_out/var_$TAG.txt
  : > $LOG
  for r in 1 2; do for name in "$@"; do
    L=$OUT/_ssfm_$name.so
    [ "$name" = product ] && L=$ROOT/opticomlib_amd/_ssfm_amd.so
    echo -n "$name single: " >> $LOG
    SSFM_LIB=$L python $ROOT/bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f us/step'%d['us_per_ssfm_step'], {k: round(v,2) for k,v in r['launch_us'].items()})" >> $LOG 2>&1
    if [ "${FOUR:-1}" = 1 ]; then
    echo -n "$name 4 fields: " >> $LOG
    SSFM_LIB=$L python $ROOT/tools/four_fields.py >> $LOG 2>&1
    fi
  done; done
  cat $LOG
fi
