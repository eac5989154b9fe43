#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes (one counter per pass, as MI355X_MICROARCH.md prescribes) per kernel family.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced
reads, so the read side is doubled (guide section "HBM").  traffic = (2*FETCH + WRITE) * 1024 / launches.

    python tools/pmc_summary.py --mfma <counter_collection.csv> <out.json>
summarises a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass: matrix-core busy share per
kernel family.  SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16 (checked here against the instruction count
the contraction sizes imply); busy share = busy / (dispatch time x 1024 SIMDs x clock), quoted at the nominal 2.4 GHz and at
the clock GRBM_GUI_ACTIVE / 8 XCDs reports for the same dispatches."""
import collections
import csv
import json
import sys


def fam(name: str) -> str:
    if "gemm_bf16s_kernel<1" in name:
        return "gemm_bf16x1_kernel"
    if "gemm_bf16s_kernel<2" in name:  # the last template argument tells the fp16 pieces (f16x3: NAR) from the bf16 ones (bf16x3: Mimi)
        return "gemm_f16x3_kernel" if name.split(">(")[0].rstrip().endswith("true") else "gemm_bf16x3_kernel"
    if "gemm_8p_kernel" in name:  # round 6: the long-K form of the three-pass contraction (csrc/gemm_8p.hip) belongs to the decoder's family
        return "gemm_bf16x3_kernel"
    if "gemm_bf16s_kernel<3" in name:
        return "gemm_bf16x6_kernel"
    if "attn_mfma_split_kernel" in name:
        return "attention_split_kernel"
    if "attn_window_mfma_kernel" in name or "attn_mfma_kernel" in name:
        return "attention_kernel"
    if "seanet_tail" in name:  # (the four-wave and the sixteen-wave kernel)
        return "seanet_tail_kernel"
    for k in ("gemm_f32_kernel", "skinny_kernel", "xattn_step_kernel", "ar_sample_kernel", "seanet_uptail_kernel", "seanet_tail_kernel", "seanet_res128_kernel", "seanet_up128_kernel",
              "attention_kernel", "argmax_partials_kernel"):
        if k in name:
            return k
    return "other"


def mfma(path, out_path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        d[f][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            d[f]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d[f]["n"] += 1
    out = {}
    for f, v in d.items():
        busy, ns, gui = v["SQ_VALU_MFMA_BUSY_CYCLES"], v["ns"], v["GRBM_GUI_ACTIVE"] / 8.0
        if busy <= 0:
            continue
        out[f] = {"launches": int(v["n"]), "dispatch_ms": round(ns / 1e6, 3), "mfma_busy_cycles": round(busy),
                  "mfma_busy_cycles_per_launch": round(busy / v["n"]), "clock_ghz_from_gui_active": round(gui / ns, 3),
                  "mfma_busy_share_at_2p4ghz": round(busy / (ns * 2.4 * 1024), 4), "mfma_busy_share_at_measured_clock": round(busy / (gui * 1024), 4)}
    import os

    json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- " +
                          os.environ.get("PMC_COMMAND", "python bench.py --lanes 1 --steps 1 --warmup 2 --profile-steps 0 --no-cpu-baseline --ttfa-runs 0 (3 passes of the step, whole chip per kernel)"),
               "note": "1024 = 256 CUs x 4 SIMDs; one v_mfma_f32_32x32x16_bf16 keeps a SIMD's matrix core busy for 32 cycles, v_mfma_f32_32x32x2_f32 for 64",
               "families": out}, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if len(sys.argv) > 1 and sys.argv[1] == "--mfma":
    mfma(sys.argv[2], sys.argv[3])
    sys.exit(0)


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        d[f][0] += 1
        d[f][1] += float(r["Counter_Value"])
    return d


import os

f, w = agg(sys.argv[1]), agg(sys.argv[2])
out = {}
for k in f:
    n = f[k][0]
    out[k] = {"launches": n, "fetch_kib_raw": round(f[k][1]), "write_kib": round(w.get(k, [0, 0])[1]),
              "traffic_bytes_per_launch": round((2 * f[k][1] + w.get(k, [0, 0])[1]) * 1024 / max(1, n))}
# PMC_ROWS / PMC_PRECISION / PMC_COMMAND: set by tools/collect_evidence.sh to what the profiled command ran (rows of an AR frame)
json.dump({"command": os.environ.get("PMC_COMMAND", "python bench.py --lanes 1 --steps 1 --warmup 2 --profile-steps 0 --no-cpu-baseline --ttfa-runs 0 (3 passes of the step)"),
           "ar_rows_per_frame": int(os.environ.get("PMC_ROWS", "32")), "precision": os.environ.get("PMC_PRECISION", "f32"),
           "correction": "read side doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane reads)", "families": out},
          open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
