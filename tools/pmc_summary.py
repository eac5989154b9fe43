#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes (one counter per pass, as MI355X_MICROARCH.md prescribes) per kernel family.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced
reads, so the read side is doubled (guide section "HBM").  traffic = (2*FETCH + WRITE) * 1024 / launches."""
import collections
import csv
import json
import sys


def fam(name: str) -> str:
    if "gemm_bf16s_kernel<2" in name:
        return "gemm_bf16x3_kernel"
    if "gemm_bf16s_kernel<3" in name:
        return "gemm_bf16x6_kernel"
    if "attn_window_mfma_kernel" in name:
        return "attention_kernel"
    for k in ("gemm_f32_kernel", "skinny_kernel", "xattn_step_kernel", "ar_sample_kernel", "seanet_tail_kernel", "attention_kernel"):
        if k in name:
            return k
    return "other"


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        d[f][0] += 1
        d[f][1] += float(r["Counter_Value"])
    return d


f, w = agg(sys.argv[1]), agg(sys.argv[2])
out = {}
for k in f:
    n = f[k][0]
    out[k] = {"launches": n, "fetch_kib_raw": round(f[k][1]), "write_kib": round(w.get(k, [0, 0])[1]),
              "traffic_bytes_per_launch": round((2 * f[k][1] + w.get(k, [0, 0])[1]) * 1024 / max(1, n))}
json.dump({"command": "python bench.py --lanes 1 --steps 1 --warmup 2 --profile-steps 0 --no-cpu-baseline --ttfa-runs 0 (3 passes of the step)",
           "correction": "read side doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane reads)", "families": out},
          open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
