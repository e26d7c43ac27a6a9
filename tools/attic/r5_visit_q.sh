#!/bin/bash
# round 5, a diagnostic visit: where a capture run's time goes (parts switched off one at a time), and every workgroup's line of the one-launch filter's timeline
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
for b in 0 1 3 35 16 4 0; do timeout 300 python3 tools/attic/capture_ab.py $b 3; done
} > $O/r5q_capture_ab.txt 2>&1
{
SSFM_LIB=$R/build/var/_ssfm_tl.so SOS_TL_PATH=$O/r5q_sos_tl_full.txt timeout 300 python3 tools/sos_timeline.py
SSFM_LIB=$R/build/var/_ssfm_tl.so SOS_TL_PATH=$O/r5q_sos_tl_full_b.txt timeout 300 python3 tools/sos_timeline.py
} > $O/r5q_sos_tl.txt 2>&1
tail -30 $O/r5q_capture_ab.txt
