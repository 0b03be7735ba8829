"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (siftmetal_amd/csrc/resource_usage.txt)."""
import re
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "siftmetal_amd/csrc/resource_usage.txt").read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = re.sub(r"^_ZN6siftmi", "", b.split("\n")[0].strip())[:64]
    vals = []
    for label, k in keys:
        m = re.search(k + r": (\d+)", b)
        vals.append("%s %s" % (label, m.group(1) if m else "?"))
    print("%-66s %s" % (name, "  ".join(vals)))
