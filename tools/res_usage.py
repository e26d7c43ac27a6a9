"""Registers / occupancy of the hot kernels from a `hipcc -Rpass-analysis=kernel-resource-usage` log (stdin or file)."""
import re, subprocess, sys
t = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
pat = sys.argv[2] if len(sys.argv) > 2 else "float"
for m in re.finditer(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    d = d.replace("void ssfm::", "").split("(")[0]
    if re.search(pat, d):
        print(f"{d:50s} SGPR {m.group(2):>3} VGPR {m.group(3):>3} AGPR {m.group(4):>3} scratch {m.group(5):>3} occ {m.group(6)} lds {m.group(7)}")
