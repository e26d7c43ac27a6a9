#!/bin/bash
# Build timing-only ablation variants of the library (CPU box), or run them (GPU box).
# usage: tools/ablate.sh build | run
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/abl
VARIANTS="base:-DX=0 notwn:-DSSFM_ABL_NO_TWN=1 nop:-DSSFM_ABL_NO_P=1 notab:-DSSFM_ABL_NO_TAB=1 nofft:-DSSFM_ABL_NO_FFT=1 nonl:-DSSFM_ABL_NO_NL=1 memonly:-DSSFM_ABL_NO_FFT=1@-DSSFM_ABL_NO_NL=1 notables:-DSSFM_ABL_NO_TWN=1@-DSSFM_ABL_NO_TAB=1@-DSSFM_ABL_NO_P=1"
if [ "$1" = build ]; then
  mkdir -p $OUT
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr '@' ' ')
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$ROOT/include $flags -o $OUT/_ssfm_$name.so $ROOT/opticomlib_amd/csrc/ssfm_host.hip $ROOT/opticomlib_amd/csrc/sos_filter.hip 2>/dev/null; echo built $name ) &
  done; wait
else
  for r in 1 2; do for v in $VARIANTS; do
    name=${v%%:*}
    echo -n "$name lanes=${SSFM_LANES:-2}: "; SSFM_LIB=$OUT/_ssfm_$name.so python $ROOT/bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step'%d['us_per_ssfm_step'], d['roofline']['launch_us'])"
  done; done
fi
