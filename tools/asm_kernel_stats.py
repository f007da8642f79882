#!/usr/bin/env python3
"""Instruction counts of one kernel in a `hipcc -S --cuda-device-only` listing: asm_kernel_stats.py FILE.s SUBSTRING
(mangled-name substring; prints registers, scratch, and the mnemonic histogram).  Measurement tool."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^[_A-Za-z][\w$.]*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
c = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\t([a-z][a-z0-9_]+)", l)
    if m:
        c[m.group(1)] += 1
meta = {}
for l in lines[end:end + 400]:
    for k in ("next_free_vgpr", "private_segment_fixed_size", "next_free_sgpr"):
        m = re.search(r"\.amdhsa_%s (\d+)" % k, l)
        if m and k not in meta:
            meta[k] = int(m.group(1))
print(lines[start][:90], "| instructions", sum(c.values()), meta)
want = sys.argv[3:] or [k for k, _ in c.most_common(40)]
print("  " + "  ".join("%s=%d" % (k, c[k]) for k in want))
