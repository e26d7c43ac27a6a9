#!/bin/bash
# Round 6: the zero-phase filter's look-back limited to the totals that matter (SSFM_SOS_NEAR, sos_filter_impl.inc group_start_near), A/B on one box.
#   bash tools/sos_near_ab.sh   -> gpurun_out/r06_sos_near_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_sos_near_ab.txt
mkdir -p $R/gpurun_out
{ for r in 1 2 3; do
    echo "== all earlier totals (SSFM_SOS_NEAR=0), visit $r"; SSFM_SOS_NEAR=0 python3 $R/tools/filter_shapes.py
    echo "== the nearest that matter (default), visit $r"; python3 $R/tools/filter_shapes.py
  done; } > $O 2>&1
cat $O
