"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` logs: one line per kernel."""
import re
import sys

KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
for path in sys.argv[1:]:
    txt = open(path).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0]
        vals = []
        for k, pat in KEYS:
            m = re.search(pat + r": (\S+)", b)
            vals.append(f"{k}={m.group(1) if m else '?'}")
        short = name.split(" ")[0][:58]
        print(f"{short:60s} " + " ".join(vals))
