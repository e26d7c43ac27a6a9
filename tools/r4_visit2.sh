set -u
export TMPDIR=/tmp
V=build/var
mkdir -p gpurun_out
echo "== stamps rows=1"; ./build/stamp_harness 1
echo "== stamps rows=2"; ./build/stamp_harness 2
bash tools/ab.sh r04v2 2 "base:SSFM_LIB=$V/_ssfm_base.so" "pp:SSFM_LIB=$V/_ssfm_pp.so" "all:" "all_nothreads:SSFM_LANE_THREADS=0"
for t in tracebase tracemicro trace; do echo "== timeline $t"; SSFM_LIB=$V/_ssfm_$t.so python tools/trace_timeline.py 2>&1 | tail -12; done
echo "== per-WG"; SSFM_LIB=$V/_ssfm_trace.so python tools/trace_wg.py 2>&1 | tail -60
