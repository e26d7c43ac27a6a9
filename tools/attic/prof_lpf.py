import os, sys, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd.typing import gv, electrical_signal
gv(sps=16, R=32e9)
xe = electrical_signal(np.random.default_rng(1).standard_normal(4096))
oa.LPF(xe, BW=20e9).signal
pr = cProfile.Profile(); pr.enable()
for _ in range(200): y = oa.LPF(xe, BW=20e9).signal
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
