#!/bin/bash
# Build named variants of the whole library with extra -D flags (CPU box), or time them (GPU box).
#   tools/variants.sh build  name:-DFLAG=1@-DOTHER=2 ...      -> build/var/_ssfm_<name>.so
#   tools/variants.sh run TAG name ...                        -> gpurun_out/var_TAG.txt
# run: interleaved, two rounds; single field (bench.py, C2) and 4 fields resident per variant.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/var
SRC="ssfm_host.hip ssfm_medium.hip sos_filter.hip frontend.hip device_mem.hip chirpz.hip transmitter.hip prbs.hip"
mode=$1; shift
if [ "$mode" = build ]; then
  # one translation unit sees the flags (VSRC, default ssfm_host.hip): it is compiled per variant and linked with the product's other objects (build/obj)
  VSRC=${VSRC:-ssfm_host.hip}
  mkdir -p $OUT
  make -s -j8 -C $ROOT/opticomlib_amd/csrc > /dev/null || exit 1
  OTHERS=$(for f in $SRC; do [ $f = $VSRC ] || echo $ROOT/build/obj/${f%.hip}.o; done)
  n=0
  for v in "$@"; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr '@' ' ')
    ( cd $ROOT/opticomlib_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -mllvm -amdgpu-kernarg-preload-count=${PRELOAD:-16} $flags -c -o $OUT/obj_$name.o $VSRC 2>$OUT/$name.log \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/_ssfm_$name.so $OUT/obj_$name.o $OTHERS 2>>$OUT/$name.log && rm -f $OUT/obj_$name.o && echo built $name || { echo FAILED $name; tail -5 $OUT/$name.log; } ) &
    n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi
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
