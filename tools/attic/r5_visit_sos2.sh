#!/bin/bash
# round 5: the one-launch filter's poll interval (s_sleep between the rounds of the flag poll), interleaved
cd "$(dirname "$0")/../.."; R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
for r in 1 2 3; do
  for v in product sleep8 sleep32 sleep100 before; do
    echo "== $v (round $r)"
    if [ $v = product ]; then python3 tools/filter_shapes.py; else SSFM_LIB=$R/build/var/_ssfm_$v.so python3 tools/filter_shapes.py; fi
  done
done
} > $O/r5_sos_sleep.txt 2>&1
python3 - <<'PY'
import re,collections
d=collections.defaultdict(lambda: collections.defaultdict(list)); v=None
for l in open("gpurun_out/r5_sos_sleep.txt"):
    m=re.match(r"== (\S+)",l)
    if m: v=m.group(1); continue
    m=re.match(r"(n=\S+ rows=\d+ complex=\w+):.*kernels\s+([\d.]+) us",l)
    if m: d[m.group(1)][v].append(float(m.group(2)))
for shape,vs in d.items():
    print(shape, "  ".join(f"{k} {min(x):.1f}" for k,x in vs.items()))
PY
