#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== eager"; python tools/small_n.py 2>&1 | tail -18
echo "== SSFM_GRAPH=1"; SSFM_GRAPH=1 python tools/small_n.py 2>&1 | tail -18
