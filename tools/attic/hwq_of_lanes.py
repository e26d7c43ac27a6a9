"""Hardware queue of every k_time / k_freq dispatch after PROBE-MARK in an AMD_LOG_LEVEL=4 log of order_dependence_probe.py."""
import re, sys, collections
seen = False
per = collections.Counter()
order = []
for line in open(sys.argv[1], errors="replace"):
    if "PROBE-MARK" in line:
        seen = True
    if not seen:
        continue
    m = re.search(r"ShaderName\s*:\s*(\S+)", line)
    if m:
        name = m.group(1)
        last = name
    m = re.search(r"HWq=(0x[0-9a-f]+)", line)
    if m:
        hwq = m.group(1)
        sw = re.search(r"SWq=(0x[0-9a-f]+)", line)
        order.append((sw.group(1) if sw else "?", hwq))
        per[(sw.group(1) if sw else "?", hwq)] += 1
print("dispatches per (software queue, hardware queue) after the mark:")
for k, v in per.most_common(12):
    print("  ", k, v)
