#!/usr/bin/env python
"""What runs PER TIMED STEP: the difference of two rocprofv3 kernel-stats CSVs of `bench.py` runs that differ only in the number of
timed steps.  Set-up work (engine build, voice preparation, warm-up, graph recording) cancels; what is left is the timed path.
Usage: python tools/trace_delta.py short_kernel_stats.csv long_kernel_stats.csv <extra timed steps> [out.json]"""
import csv, json, re, sys

def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d

a, b, dsteps = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    if cb != ca:
        rows.append((k, (cb - ca) / dsteps, (tb - ta) / dsteps / 1e3))
tot = sum(r[2] for r in rows)
torch_rows = [r for r in rows if "at::native" in r[0] or "rocclr" in r[0] or r[0].startswith("void at::")]
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:72]
out = {"extra_timed_steps": dsteps, "kernel_us_per_step": round(tot, 1),
       "torch_or_runtime_kernels_per_step": {short(r[0]): {"launches": round(r[1], 2), "us": round(r[2], 2)} for r in torch_rows},
       "torch_or_runtime_us_per_step": round(sum(r[2] for r in torch_rows), 2),
       "families_per_step": {short(r[0]): {"launches": round(r[1], 2), "us": round(r[2], 1)} for r in sorted(rows, key=lambda r: -r[2])[:40]}}
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
print("kernel time per timed step: %.1f us; torch / runtime kernels per step: %d launches, %.2f us" % (tot, round(sum(r[1] for r in torch_rows)), out["torch_or_runtime_us_per_step"]))
for r in torch_rows:
    print("   ", short(r[0]), round(r[1], 2), "launches", round(r[2], 2), "us")
for r in sorted(rows, key=lambda r: -r[2])[:24]:
    print("%-72s %7.2f launches %9.1f us per step" % (short(r[0]), r[1], r[2]))
