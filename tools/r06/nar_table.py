#!/usr/bin/env python
"""r06: per-launch table of ONE refinement pass from a rocprofv3 kernel trace of tools/r06/nar_run.py (launch order, the first stage in
full, then the aggregate).   python tools/r06/nar_table.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:72] if m else n[:72]


names = [short(r["Kernel_Name"]) for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("nar_seed")]
lo, hi = marks[-2], marks[-1]
seq = rows[lo:hi]
tot, agg = 0.0, {}
for k, r in enumerate(seq):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    n = short(r["Kernel_Name"])
    if k < 40:
        print(f"{us:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '')):>9}  {n}")
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += us
print(f"--- {len(seq)} launches, {tot / 1e3:.3f} ms of kernel time")
for n, (c, us) in sorted(agg.items(), key=lambda t: -t[1][1]):
    print(f"{us:9.1f} us  x{c:3d}  {us / c:8.1f} each  {n}")
