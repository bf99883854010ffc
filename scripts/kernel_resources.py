#!/usr/bin/env python
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel whose name contains argv[2]."""
import re
import sys
t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print(name[:64], "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "spill", g("VGPRs Spill"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "SGPR", g("TotalSGPRs"), "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
