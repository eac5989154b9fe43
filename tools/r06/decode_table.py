#!/usr/bin/env python
"""r06: per-launch table of ONE decode from a rocprofv3 kernel trace of tools/r06/decode_run.py: the kernels of the last decode in
launch order with their durations.   python tools/r06/decode_table.py <kernel_trace.csv> [launches per decode]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:64] if m else n[:64]


names = [short(r["Kernel_Name"]) for r in rows]
# one decode = the span between two consecutive launches of the first kernel of the sequence (the codebook sum follows two fills)
marks = [i for i, n in enumerate(names) if n.startswith("codebook_sum")]
starts = marks[0::2]
lo, hi = starts[-2], starts[-1]
seq = rows[lo - 3:hi - 3] if lo >= 3 else rows[lo:hi]
tot = 0.0
agg = {}
for r in seq:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    n = short(r["Kernel_Name"])
    g = r.get("Grid_Size_X", r.get("Grid_Size", ""))
    print(f"{us:9.1f} us  grid {g:>9}  {n}")
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += us
print(f"--- {len(seq)} launches, {tot / 1e3:.3f} ms of kernel time")
for n, (c, us) in sorted(agg.items(), key=lambda t: -t[1][1]):
    print(f"{us:9.1f} us  x{c:3d}  {n}")
