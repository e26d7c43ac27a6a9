#!/usr/bin/env python3
"""Instruction mix of selected kernels from a hipcc -save-temps .s file (dev aid)."""
import sys
from collections import Counter

path = sys.argv[1]
names = sys.argv[2:]
s = open(path).read()
for name in names:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    ins = []
    for l in s[i:j].split("\n"):
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    groups = Counter()
    for k, v in c.items():
        g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
             else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        groups[g] += v
    print(name, len(ins), dict(groups))
    print("  ", c.most_common(28))
