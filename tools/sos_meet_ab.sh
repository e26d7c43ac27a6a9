#!/bin/bash
# Round 6: the zero-phase filter with ONE meeting of the workgroups instead of two (SSFM_SOS_MEET, sos_filter_impl.inc meet_states), A/B on one box.
#   bash tools/sos_meet_ab.sh   -> gpurun_out/r06_sos_meet_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_sos_meet_ab.txt
mkdir -p $R/gpurun_out
{ for r in 1 2 3; do
    echo "== two meetings (default), visit $r"; python3 $R/tools/filter_shapes.py
    echo "== one meeting (SSFM_SOS_MEET=1), visit $r"; SSFM_SOS_MEET=1 python3 $R/tools/filter_shapes.py
  done; } > $O 2>&1
cat $O
