#!/usr/bin/env python3
"""Per-kernel averages of a `rocprofv3 --pmc ...` pass:  sq_summary.py DIR label=regex [label=regex ...]
(every dispatch whose kernel name matches the regex counts for the label; counters are averaged per dispatch)."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
pats = [(a.split("=", 1)[0], re.compile(a.split("=", 1)[1])) for a in sys.argv[2:]]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for label, pat in pats:
            if pat.search(r["Kernel_Name"]):
                acc[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
for label, c in acc.items():
    n = len(next(iter(c.values())))
    print(label, "dispatches", n, {k: round(sum(v) / len(v)) for k, v in sorted(c.items())})
    w = c.get("SQ_WAVE_CYCLES")
    if w:
        tot = sum(w) / len(w)
        print("   fractions of SQ_WAVE_CYCLES:", {k: round(sum(v) / len(v) / tot, 3) for k, v in sorted(c.items()) if k != "SQ_WAVE_CYCLES"})
