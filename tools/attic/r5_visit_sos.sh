#!/bin/bash
# round 5: the one-launch filter after the hand-over change -- its tests, the shapes, the timeline (every workgroup's line), interleaved with the library before it
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sos or filt or LPF or BPF or lpf or bpf or PD or EDFA" 2>&1 | tail -4
{
for r in 1 2 3; do
  for v in new tag1 old; do
    true
    echo "== $v (round $r)"; if [ $v = old ]; then SSFM_LIB=$R/build/var/_ssfm_before.so python3 tools/filter_shapes.py; elif [ $v = tag1 ]; then SSFM_LIB=$R/build/var/_ssfm_tag1.so python3 tools/filter_shapes.py; else python3 tools/filter_shapes.py; fi
  done
done
} > $O/r5_sos_shapes.txt 2>&1
{ echo "== 2^20 x 2 complex128"; SSFM_LIB=$R/build/var/_ssfm_tl.so SOS_TL_PATH=$O/r5_sos_tl_full.txt python3 tools/sos_timeline.py
  echo "== 2^16 real"; SSFM_LIB=$R/build/var/_ssfm_tl.so LOG2N=16 ROWS=1 CPLX=0 python3 tools/sos_timeline.py; } > $O/r5_sos_timeline.txt 2>&1
python3 tests/diag/fuzz_filters.py > $O/r5_sos_fuzz.txt 2>&1
tail -18 $O/r5_sos_shapes.txt; head -16 $O/r5_sos_timeline.txt; tail -4 $O/r5_sos_fuzz.txt
