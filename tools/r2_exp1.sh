#!/bin/bash
# round 2, experiment 1: where does a step's time go (timing-only ablations; results of ablated builds are wrong by design)
cd $GRAFT_REPO_ROOT
V="base notwn nop notab notables nofft nonl memonly memonly_notables twnc"
tools/variants.sh run abl2 $V > /dev/null
mv gpurun_out/var_abl2.txt gpurun_out/r2_abl_lanes2.txt
FOUR=0 SSFM_LANES=1 tools/variants.sh run abl1 $V > /dev/null
mv gpurun_out/var_abl1.txt gpurun_out/r2_abl_lanes1.txt
echo "== lanes 2"; cat gpurun_out/r2_abl_lanes2.txt; echo "== lanes 1"; cat gpurun_out/r2_abl_lanes1.txt
