"""Registers / scratch / occupancy of the zero-phase filter's one-launch kernels from a `hipcc -Rpass-analysis=kernel-resource-usage` log."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "<2, 2"
for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    k = re.search(r"(chunk_\w+::k_\w+<[^>]*>)", d)
    if k and re.search(pat, k.group(1)):
        print(f"{k.group(1):45s} SGPR {m.group(2):>3} VGPR {m.group(3):>3} scratch {m.group(4):>3} occ {m.group(5)} lds {m.group(6)}")
