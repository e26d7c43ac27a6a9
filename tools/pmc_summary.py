#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per kernel -> profiles/pmc_traffic.json.

usage: pmc_summary.py <dir with *counter_collection.csv from the FETCH pass> <dir of the WRITE pass> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB (guide: bytes = value * 1024).  gfx950: FETCH_SIZE under-counts wide
(16 B/lane) coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); narrower accesses are uncalibrated, so
the raw value AND the x2-corrected value are both recorded, together with dispatches of known byte counts
(the D2D field copy) for calibration.  With a 56 MiB working set these counters see L2<->Infinity-Cache
fabric requests, not DRAM."""
import csv, glob, json, os, sys
from collections import defaultdict

def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc

def short(name):
    if "k_time" in name: return "k_time"
    if "k_freq" in name: return "k_freq"
    if "copyBuffer" in name: return "copyBuffer"
    return None

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_summary.py"}
for name in sorted(set(fetch) | set(write)):
    s = short(name)
    if s is None: continue
    f = fetch.get(name, []); w = write.get(name, [])
    fm = sum(f) / len(f) * 1024 if f else None
    wm = sum(w) / len(w) * 1024 if w else None
    e = out.setdefault(s, {"kernels": []})
    e["kernels"].append({"name": name[:80], "dispatches": len(f), "fetch_bytes_raw": fm, "write_bytes": wm})
for s, e in out.items():
    if s.startswith("_"): continue
    ks = [k for k in e["kernels"] if k["dispatches"] > 10] or e["kernels"]
    k = max(ks, key=lambda k: k["dispatches"])
    e["fetch_bytes_raw"] = k["fetch_bytes_raw"]; e["write_bytes"] = k["write_bytes"]
    if k["fetch_bytes_raw"] is not None and k["write_bytes"] is not None:
        e["bytes_per_launch_raw"] = k["fetch_bytes_raw"] + k["write_bytes"]
        e["bytes_per_launch"] = 2 * k["fetch_bytes_raw"] + k["write_bytes"]      # gfx950 FETCH_SIZE x2 correction
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
