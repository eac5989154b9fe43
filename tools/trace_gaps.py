#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> per kernel of the AR frame: average duration and the average gap to the NEXT kernel on the
same queue (start[i+1] - end[i]); the boundary cost the frame pays 23 times.  python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
by_q = defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for q, ks in by_q.items():
    ks.sort()
    for i, (s, e, n) in enumerate(ks):
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        if not any(k in short for k in ("skinny_kernel", "xattn_step", "ar_sample")):
            continue
        if i + 1 < len(ks) and ks[i + 1][0] - e < 50_000:  # the same frame chain (not the idle time between phases)
            dur[short] += e - s
            gap[short] += ks[i + 1][0] - e
            cnt[short] += 1
tot_d = tot_g = tot_n = 0
for k in sorted(cnt, key=lambda k: -cnt[k]):
    print(f"{k:62s} n={cnt[k]:6d} avg kernel {dur[k] / cnt[k] / 1e3:6.2f} us  avg gap to next {gap[k] / cnt[k] / 1e3:6.2f} us")
    tot_d += dur[k]; tot_g += gap[k]; tot_n += cnt[k]
if tot_n:
    print(f"all AR kernels: n={tot_n} avg kernel {tot_d / tot_n / 1e3:.2f} us, avg gap {tot_g / tot_n / 1e3:.2f} us  (x23 per frame: {23 * tot_d / tot_n / 1e3:.1f} + {23 * tot_g / tot_n / 1e3:.1f} us)")
