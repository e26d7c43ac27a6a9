import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from opticomlib_amd import _lib
for prec, name in ((_lib.C128, "c128"), (_lib.C64, "c64")):
    for k in (20, 21, 22):
        for rep in range(2):
            t = time.perf_counter(); p = _lib.Plan(1 << k, 2, prec); p.synchronize(); el = time.perf_counter() - t
            info = p.last_run_info() if hasattr(p, "last_run_info") else None
            print(f"{name} 2^{k} x 2 plan #{rep}: {el*1e3:.1f} ms", {k_: getattr(info, k_) for k_ in ("lanes", "lanes_from_pool", "lane_ratings_total", "lane_pairs_reused", "lanes_dropped", "lanes_remade") if info is not None and hasattr(info, k_)}, flush=True)
            p.close()
