#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats CSV -> the per-kernel table of the AR frame that bench.py quotes in its roofline entry.
    python tools/ar_kernel_table.py <kernel_stats.csv> <out.json> [frames_in_the_run]
Average duration per launch of every kernel of the frame graph (skinny instantiations, cross-attention step, sampler),
launches per frame, and their sum = the frame's kernel time."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out, per_frame = {}, {}
fam_of = {"skinny_kernel": "skinny_kernel", "xattn_step_kernel": "xattn_step_kernel", "ar_sample_kernel": "ar_sample_kernel"}
samp = [r for r in rows if "ar_sample_kernel" in r["Name"]]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else (int(samp[0]["Calls"]) if samp else 0)
total = 0.0
for r in rows:
    for k, fam in fam_of.items():
        if k in r["Name"]:
            short = r["Name"].split("::")[-1].split("(")[0]
            calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
            n = round(calls / max(1, frames))
            out[short] = {"avg_us": round(avg, 2), "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
                          "launches_per_frame": n}
            per_frame[fam] = per_frame.get(fam, 0) + n
            total += avg * n
json.dump({"source": sys.argv[1], "frames": frames, "kernels": out, "launches_per_frame": per_frame,
           "frame_kernel_time_us": round(total, 2)}, open(sys.argv[2], "w"), indent=1)
print(json.dumps({"frame_kernel_time_us": round(total, 2), "launches_per_frame": per_frame, "kernels": out}, indent=1))
