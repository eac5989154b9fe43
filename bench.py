#!/usr/bin/env python
"""Benchmark of the Sopro synthesize hot path on MI355X (BASELINE.json metric: audio-seconds per
wall-second, plus p50 time-to-first-audio of stream()).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input: 32 utterances x 200 frames
(BASELINE.json configs[1]) through conditioning -> AR loop (hipGraph replay) -> NAR refinement -> Mimi
decode, with weights, text ids and the prepared reference voice already resident in HBM when the timed
region starts.  N > 1 (launched by torch.distributed.run, one rank per GPU) is weak scaling: every rank
synthesises its own 32 utterances, there is no data-path collective; the timed region is bracketed by a
barrier + device sync on both sides and the MAX over ranks is reported.

One JSON line is printed by rank 0.  Extra objects on it:
  roofline     -- the dominant kernel family of the step BY SUMMED HIP-EVENT TIME over every instrumented family, the AR
                  frame graph included (it is the largest: the 23-launch per-frame graph of the autoregressive loop):
                  algorithmic bytes (HBM bound) or flops (MFMA bound) per launch / average launch time; the other
                  families follow in roofline_more
  parity       -- ties the timed run to correct results: every timed step ran the same seeded job, so the first and the
                  last step must be bit-identical; and one greedy 32 x 200 batch is generated next to the timed region and
                  row 0 is compared with the CPU oracle in the baseline child (codebook 0 exact, refined tokens audited)
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, validated against the reference
                  in tests/) timed on this host on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BATCH = 32
FRAMES = 200          # max_frames = 199 -> 200 AR steps (reference runs max_frames + 1 steps, model.py:242)
TEXT_LEN = 64
REF_FRAMES = 150
VOCAB = 512           # synthetic text table (the real 128k-row table only changes a gather)
FRAME_SEC = 0.08
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


class Tok:
    vocab_size = VOCAB

    def encode(self, text):
        raise RuntimeError("bench passes token ids directly")


def build_engine(device: str, precision: str = "f32"):
    from sopro_amd import SoproTTS
    from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
    from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    wn = synth_sopro_weights(cfg, VOCAB, 0, suppress_eos=True)  # EOS bias -1e9: fixed-length runs
    mn = synth_mimi_weights(mc, 0)
    return SoproTTS.from_weights(cfg, wn, mn, Tok(), device=device, precision=precision), cfg, mc, wn, mn


def make_inputs(rank: int):
    rng = np.random.default_rng(1000 + rank)
    ids = [torch.from_numpy(rng.integers(0, VOCAB, size=TEXT_LEN)) for _ in range(BATCH)]
    ref_tq = torch.from_numpy(rng.integers(0, 2048, size=(REF_FRAMES, 32)))
    return ids, ref_tq


def make_voices(rank: int, n: int):
    """One reference voice per utterance (ref_tokens of utterance i from seed 1000 + index).  Voice 0 is the
    voice of make_inputs(rank), so row 0 of a batch is the utterance the oracle leg checks."""
    out = [make_inputs(rank)[1]]
    for i in range(1, n):
        rng = np.random.default_rng(1000 + 4096 * (rank + 1) + i)
        out.append(torch.from_numpy(rng.integers(0, 2048, size=(REF_FRAMES, 32))))
    return out


def oracle_parity(cfg, wn, path: str):
    """Row 0 of the engine's greedy batch (token matrix saved by the parent) against the oracle: codebook 0 must be
    equal; refined tokens equal, or every deviating decision an audited near-tie (oracle.nar_audit)."""
    from oracle import sopro_oracle as O

    got = torch.from_numpy(np.load(path)).long()
    w = O.to_torch(wn)
    ids, ref_tq = make_inputs(0)
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    with torch.inference_mode():
        ref = O.prepare_reference(ref_tq, w, cfg)
        kw = dict(max_frames=FRAMES - 1, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=float(cfg.style_strength))
        want = O.generate_tokens(ids[0], ref, w, cfg, **kw)
        out = {"row": 0, "frames": int(want.shape[0]), "shape_equal": tuple(want.shape) == tuple(got.shape)}
        if out["shape_equal"]:
            out["codebook0_equal"] = bool(torch.equal(want[:, 0], got[:, 0]))
            out["refined_mismatches"] = int((want != got).sum())
            if out["refined_mismatches"] and out["codebook0_equal"]:
                prep = O.prepare_conditioning(ids[0], ref, w, cfg, max_frames=FRAMES - 1, style_strength=kw["style_strength"])
                n_off, gap = O.nar_audit(prep["cond_ar"][:, : got.shape[0]], got.unsqueeze(0), w, cfg)
                out["audit_off_argmax"], out["audit_worst_logit_gap"] = n_off, gap
        out["ok"] = bool(out["shape_equal"] and out.get("codebook0_equal") and
                         (out["refined_mismatches"] == 0 or out.get("audit_worst_logit_gap", 1.0) < 1e-4))
    return out


def cpu_baseline(cfg, mc, wn, mn, n_utts: int):
    """The oracle on the host cores, one utterance at a time (the reference has no batched API)."""
    from oracle import sopro_oracle as O

    w, mw = O.to_torch(wn), O.to_torch(mn)
    ids, ref_tq = make_inputs(0)
    # the reference's per-frame loop is dispatch-bound and stops scaling at ~4 threads;
    # more intra-op threads only add synchronisation cost, so the baseline uses min(host cores, 8)
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    with torch.inference_mode():
        ref = O.prepare_reference(ref_tq, w, cfg)
        O.synthesize(ids[0], ref, w, mw, cfg, mc, max_frames=7, top_p=0.9, temperature=1.05, anti_loop=True)  # warm-up
        t0 = time.perf_counter()
        frames = 0
        for i in range(n_utts):
            wav = O.synthesize(ids[i % len(ids)], ref, w, mw, cfg, mc, max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True)
            frames += wav.shape[-1] // 1920
        dt = time.perf_counter() - t0
        # time-to-first-audio of the oracle's stream() (reference: src/sopro/streaming.py:24-130), same arguments as the GPU leg
        lat = []
        n_ttfa = int(os.environ.get("SOPRO_BENCH_CPU_TTFA_RUNS", "20"))
        for i in range(n_ttfa + 2):
            t1 = time.perf_counter()
            it = O.stream(ids[i % len(ids)], ref, w, mw, cfg, mc, max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, chunk_frames=6)
            next(it)
            if i >= 2:
                lat.append((time.perf_counter() - t1) * 1e3)
            it.close()
    out = {"value": round(frames * FRAME_SEC / dt, 3), "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_utts} utterances x {FRAMES} frames, sequential synthesize() of oracle/sopro_oracle.py (torch fp32 CPU), "
                     f"{dt:.1f} s wall"}
    if lat:
        out["ttfa_ms_p50"] = round(float(np.percentile(lat, 50)), 2)
        out["ttfa_runs"] = len(lat)
        out["ttfa_what"] = "first chunk of oracle stream(chunk_frames=6), same text / voice / sampling arguments as ttfa_ms_p50 of the GPU leg"
    return out


def ar_step_bytes(B: int, S: int, wbytes: int = 4, abytes: int = 4) -> float:
    """weights + cond/embedding rows + ring buffers + K/V per AR frame (w = a = 4 bytes in fp32)."""
    return 10_575_492 * wbytes + B * (768 + 32_256 + 2304 * S) * abytes


def latest_profile(suffix: str) -> dict:
    """profiles/rNN_<suffix> of the latest round that has one (rocprofv3 summaries of this same command, tools/pmc_summary.py,
    tools/ar_kernel_table.py); {} when there is none."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        d.setdefault("source", os.path.relpath(files[-1], ROOT))
        return d
    except Exception:  # noqa: BLE001
        return {}


def profile_stamp(path: str) -> dict:
    """Identity of a committed profile file whose numbers are REPLAYED into the line (not measured by this run): path +
    content hash (there is no .git on the GPU box; `git log -- <file>` on the same content gives the commit)."""
    import hashlib

    try:
        return {"file": os.path.relpath(path, ROOT), "sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16], "replayed": True}
    except OSError:
        return {"file": path, "replayed": True}


def rocprof_family_table() -> dict:
    """Per-launch kernel time of the instrumented families from the latest committed rocprofv3 --kernel-trace --stats summary of
    this command (profiles/rNN_bench_lanes4_kernel_stats.csv): kernel time proper, where HIP events around a sub-50-us launch on
    a shared partition mostly measure queueing."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_lanes4_kernel_stats.csv")))
    if not files:
        return {}
    pat = {"gemm_f32_kernel": "gemm_f32_kernel", "gemm_bf16s_kernel<1,": "gemm_bf16x1_kernel", "gemm_bf16s_kernel<2,": "gemm_bf16x3_kernel",
           "gemm_bf16s_kernel<3,": "gemm_bf16x6_kernel", "gemm_8p_kernel": "gemm_bf16x3_kernel",  # (the long-K form: the decoder's family)  # (<2, ..., true> = the fp16 pieces of the NAR path: told apart below)
           "attention_kernel": "attention_kernel", "attn_window_mfma_kernel": "attention_kernel", "attn_mfma_kernel": "attention_kernel",
           "attn_decode_kernel": "attention_kernel", "attn_mfma_split_kernel": "attention_split_kernel", "seanet_tail_kernel": "seanet_tail_kernel", "seanet_tail16_kernel": "seanet_tail_kernel", "seanet_res128_kernel": "seanet_res128_kernel",
           "seanet_up128_kernel": "seanet_up128_kernel", "seanet_uptail_kernel": "seanet_uptail_kernel"}
    out: dict = {}
    try:
        for r in csv.DictReader(open(files[-1])):
            for k, fam in pat.items():
                if k in r["Name"]:
                    if fam == "gemm_bf16x3_kernel" and r["Name"].split(">(")[0].rstrip().endswith("true"):
                        fam = "gemm_f16x3_kernel"
                    d = out.setdefault(fam, {"calls": 0, "ns": 0.0})
                    d["calls"] += int(r["Calls"])
                    d["ns"] += float(r["TotalDurationNs"])
                    break
    except Exception:  # noqa: BLE001
        return {}
    return {"families": {k: round(v["ns"] / max(1, v["calls"]) / 1e3, 2) for k, v in out.items()}, "stamp": profile_stamp(files[-1])}


def run_leg(tts, ids, refs, *, frames: int, steps: int, lanes: int, args, seed: int = 20260924) -> dict:
    """A short throughput run of another shape / voice set / engine right after the headline run, on the same box: warm-up
    (eager pass + recording of the launch sequences), then `steps` timed steps bracketed by device syncs."""
    B = len(ids)
    job = dict(texts=[""] * B, refs=list(refs), max_frames=frames - 1, top_p=0.9, temperature=1.05, anti_loop=True, text_ids=list(ids), seed=seed)
    pipe = None
    if lanes > 1:
        from sopro_amd.pipeline import PipelinedSynthesizer

        pipe = PipelinedSynthesizer(tts, lanes=lanes, ar_cus=args.ar_cus, ar_parts=args.ar_parts, ar_shared=bool(args.ar_shared), bulk_slots=args.bulk_slots)

    phase_s: dict = {}
    co = coalesce_arg(args.coalesce)
    if pipe is not None:  # every lane records every pass shape of the timed run before anything is timed
        pipe.prepare(job, sizes=set(pipe.pass_sizes(steps, co)))

    def go(n, timings=None):
        outs = pipe.run([job] * n, coalesce=co) if pipe is not None else [tts.synthesize_batch(timings=timings, **job) for _ in range(n)]
        for out in outs:
            assert all(o.shape[-1] == frames * 1920 for o in out)

    try:
        go(max(2, lanes if pipe is not None else 2))
        torch.cuda.synchronize()
        c0, t0 = time.process_time(), time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        if pipe is None:  # sequential leg: one more repeat with the phase timers on (they synchronise between the phases)
            go(steps, phase_s)
            torch.cuda.synchronize()
    finally:
        if pipe is not None:
            pipe.close()
    out = {"value": round(steps * B * frames * FRAME_SEC / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "batch": B, "frames": frames, "lanes": lanes, "voices": len({id(r) for r in refs}), "host_cpu_s_per_step": round(cpu / steps, 4)}
    if phase_s:
        out["phase_ms_per_step"] = {k: round(v / steps * 1e3, 3) for k, v in phase_s.items() if not k.startswith("_")}
    return out


def coalesce_arg(v):
    """--coalesce: an integer (jobs per pass) or "auto" (PipelinedSynthesizer.pass_sizes: 2 per pass for short queues, 4 from 4 queued jobs per lane)."""
    if str(v) == "auto":
        return "auto"
    if "," in str(v):  # explicit pass sizes, e.g. "4,4,6,6" (developer sweeps)
        return [max(1, int(x)) for x in str(v).split(",")]
    return max(1, int(v))


def bf16_quality(tts, tts16, ids, refs, frames: int, cfg, n: int = 4) -> dict:
    """What the bf16 mode costs in quality, measured against the fp32 engine on the same inputs (greedy decoding, n utterances):
    how long free-running codebook-0 generation stays identical, how many refined tokens agree given the fp32 engine's
    codebook 0 and conditioning, and the decoder's waveform error on the fp32 engine's tokens.  (tests/test_gpu_bf16_mode.py
    measures the same things against the fp32 REFERENCE fixtures.)"""
    kw = dict(max_frames=frames - 1, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=float(cfg.style_strength))
    g32 = tts.model.generate_tokens_batch(ids[:n], refs[:n], **kw)
    g16 = tts16.model.generate_tokens_batch(ids[:n], refs[:n], **kw)
    first = []
    for a, b in zip(g32, g16):
        m = min(int(a.shape[0]), int(b.shape[0]))
        neq = (a[:m, 0] != b[:m, 0]).nonzero()
        first.append(int(neq[0]) if neq.numel() else m)
    T = min(int(g.shape[0]) for g in g32)
    r32 = torch.stack([g[:T] for g in g32])
    prep = tts.model.prepare_conditioning_batch(ids[:n], refs[:n], max_frames=frames - 1, style_strength=kw["style_strength"])
    r16 = tts16.model.nar_refine(prep["cond_ar"][:, :T], r32[:, :, 0])
    agree = float((r16[:, :, 1:] == r32[:, :, 1:]).float().mean())
    w32, w16 = tts.codec.decode_batch(r32).float(), tts16.codec.decode_batch(r32).float()
    torch.cuda.synchronize()
    err = float((w16 - w32).abs().max() / w32.abs().max())
    snr = 10.0 * float(torch.log10((w32 ** 2).mean() / ((w16 - w32) ** 2).mean()))
    return {"against": "the fp32 engine on the same inputs", "utterances": n, "frames": T,
            "greedy_codebook0_identical_frames": first, "refined_token_agreement_given_fp32_codebook0": round(agree, 4),
            "waveform_max_err_of_peak": float(f"{err:.3e}"), "waveform_snr_db": round(snr, 1)}


LEGS = ("one_voice_32x200", "f32_32x400", "f32_1x400_sequential", "bf16_32x200", "bf16_32x400")
BF16_DETAIL = ("bf16 mode (round 4: bf16 IN MEMORY, not only in flight): the AR frame streams bf16 weights, bf16 folded text operands "
               "K' / V' and bf16 ring buffers; the SEANet decoder's activations are bf16 rows in memory (contractions read / write "
               "them directly, the three fused kernels have bf16-row forms); NAR + Mimi contractions one MFMA pass on bf16 operands; "
               "fp32 accumulators, norms, softmax, residual streams, the codec transformer's stream and the conditioning.  A "
               "throughput mode, not a parity mode: quote it with its quality block (gate: tests/test_gpu_bf16_mode.py)")


def leg_main(name: str, args, device: str) -> dict:
    """One extra leg of the default line in a process of its own (see main): same inputs as the headline (rank 0), another shape /
    voice set / engine."""
    bf16 = name.startswith("bf16")
    tts, cfg, _mc, _wn, _mn = build_engine(device, "bf16" if bf16 else "f32")
    ids, _ref_tq = make_inputs(0)
    voices = [tts.prepare_reference(ref_tokens_tq=v) for v in make_voices(0, BATCH)]
    if name == "one_voice_32x200":
        return run_leg(tts, ids, [voices[0]] * BATCH, frames=FRAMES, steps=args.steps, lanes=args.lanes, args=args)
    if name == "f32_32x400":
        return run_leg(tts, ids, voices, frames=400, steps=6, lanes=args.lanes, args=args)
    if name == "f32_1x400_sequential":
        out = run_leg(tts, ids[:1], voices[:1], frames=400, steps=8, lanes=1, args=args)
        # BASELINE's north star asks for batch 1 "as a fraction of the HBM roofline": the AR frame of ONE utterance against the bytes
        # it has to move (the frame's weights once + one row's state) and against the floor of a launch-per-stage design
        ar_ms = (out.get("phase_ms_per_step") or {}).get("ar")
        if ar_ms:
            us = ar_ms * 1e3 / 400.0
            bts = ar_step_bytes(1, TEXT_LEN, 4)
            out["roofline"] = {"kernel": "AR frame at batch 1 (hipGraph of 23 dependent launches), whole chip, nothing else running",
                               "bound": "hbm", "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": bts,
                               "achieved": round(bts / (us * 1e-6) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(bts / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 5),
                               "hbm_floor_us": round(bts / (PEAK_HBM_GBS * 1e9) * 1e6, 2),
                               "stage_floor_us": round(23 * 4.6, 1),
                               "frac_of_stage_floor": round(23 * 4.6 / us, 3),
                               "stage_floor_note": "23 dependent launches x 4.6 us = boundary 1.9 + body on hot operands 1.5 + the activation's trip through "
                                                   "the fabric 1.2-1.5 (profiles/r03_boundary_probe.txt); two resident-kernel designs were built and measured "
                                                   "slower at 1 / 4 / 16 / 32 rows (profiles/r03_persist_pair.txt, r04_persist_pair_small_batches.txt): the "
                                                   "batch-1 frame sits on the floor of a launch-per-stage design, which is 20x above the HBM floor"}
        return out
    if name == "bf16_32x200":
        out = run_leg(tts, ids, voices, frames=FRAMES, steps=args.steps, lanes=args.lanes, args=args)
        out["dtype"], out["dtype_detail"] = "bf16", BF16_DETAIL
        tts32 = build_engine(device, "f32")[0]
        v32 = [tts32.prepare_reference(ref_tokens_tq=v) for v in make_voices(0, 4)]
        out["quality"] = bf16_quality(tts32, tts, ids, v32, FRAMES, cfg)
        return out
    if name == "bf16_32x400":
        out = run_leg(tts, ids, voices, frames=400, steps=6, lanes=args.lanes, args=args)
        out["dtype"] = "bf16"
        return out
    raise SystemExit(f"unknown leg {name!r}")


LINE_LIMIT = 8192  # the driver parses the last stdout line from a bounded buffer (round 5: a 21 kB line came back as parsed = null)
FULL_RECORD = "bench_full.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full: dict) -> dict:
    """The ONE stdout line: headline keys, `config` (second metric + compact legs), `roofline` (primary family + the batch-1 frame),
    `roofline_more` reduced to the figures a reader checks, `cpu_baseline`, `parity`.  Everything else (per-kernel tables, prose,
    profile stamps, full legs) is in FULL_RECORD next to bench.py, in gpurun_out/ when that exists, and on stderr."""
    roof_keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "launches", "avg_launch_us", "ms_per_step",
                 "concurrent_phases", "wall_ms_per_step", "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "rows_per_launch",
                 "cu_share", "frac_of_pass_ceiling", "mfma_busy_pmc", "avg_launch_us_rocprof", "phase", "frac_reading")
    more_keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_pass_ceiling", "ms_per_step", "avg_launch_us", "traffic",
                 "mfma_busy_pmc", "phase")
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "warmup_run", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    out["vs_baseline"] = full.get("vs_baseline")
    cfg = full.get("config") or {}
    c = _pick(cfg, ("workload", "batch_per_gpu", "frames", "voices_per_batch", "parallelism", "lanes_per_gpu", "pass_sizes", "legs"))
    if cfg.get("second_metric"):
        c["second_metric"] = _pick(cfg["second_metric"], ("ttfa_ms_p50", "cpu_ttfa_ms_p50", "what"))
    out["config"] = c
    out["phase_ms_per_step"] = full.get("phase_ms_per_step")
    r = full.get("roofline")
    if r:
        rr = _pick(r, roof_keys)
        rr["kernel"] = rr["kernel"].split(" (")[0] + (" (hipGraph, 23 launches)" if rr["kernel"].startswith("AR frame") else "")
        if r.get("isolated_whole_chip"):
            rr["isolated_whole_chip"] = _pick(r["isolated_whole_chip"], ("avg_launch_us", "achieved_GBps", "frac"))
        if r.get("batch1"):
            rr["batch1"] = _pick(r["batch1"], ("bound", "avg_launch_us", "achieved", "peak", "unit", "frac", "hbm_floor_us", "stage_floor_us", "frac_of_stage_floor"))
        out["roofline"] = rr
    else:
        out["roofline"] = None
    more = []
    for e in full.get("roofline_more") or []:
        m = _pick(e, more_keys)
        m["kernel"] = m["kernel"].split(" (")[0]
        more.append(m)
    out["roofline_more"] = more
    cb = full.get("cpu_baseline")
    if cb:
        o = _pick(cb, ("value", "unit", "cores", "kind", "sample", "ttfa_ms_p50"))
        if cb.get("reference_ratio"):
            o["reference_ratio"] = _pick(cb["reference_ratio"], ("oracle_over_reference", "estimated_reference_value_here", "file"))
        out["cpu_baseline"] = o
    else:
        out["cpu_baseline"] = None
    par = full.get("parity")
    out["parity"] = _pick(par, ("ok", "row", "frames", "codebook0_equal", "refined_mismatches", "audit_worst_logit_gap", "timed_steps_identical",
                                "timed_outputs_finite", "rank_output_sha16", "f16_range_fallbacks", "mode")) if par else None
    out["full_record"] = FULL_RECORD
    return out


def emit(full: dict, root: str) -> str:
    """Write the full record to a side file (+ stderr) and print the compact line (< LINE_LIMIT bytes) as the last stdout line."""
    blob = json.dumps(full)
    for d in (root, os.path.join(root, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, FULL_RECORD), "w") as fh:
                    fh.write(blob + "\n")
            except OSError as e:  # a read-only tree must not void the measurement
                log(f"could not write {FULL_RECORD} in {d}: {e!r}")
    print("[bench full record] " + blob, file=sys.stderr, flush=True)
    text = json.dumps(compact_line(full), separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT}): trim compact_line()"
    print(text, flush=True)
    return text


_T0 = time.perf_counter()


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    global BATCH, FRAMES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--profile-steps", type=int, default=4, help="instrumented repeat of the steps for the per-kernel roofline (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=6)
    ap.add_argument("--ttfa-runs", type=int, default=50, help="stream() calls timed for the p50 time-to-first-audio (after 5 warm-ups)")
    ap.add_argument("--lanes", type=int, default=4, help="engines pipelined on one GPU (1 = strictly sequential batches)")
    ap.add_argument("--ar-cus", type=int, default=64, help="CUs of each AR partition (latency-bound phase) when lanes > 1")
    ap.add_argument("--ar-shared", type=int, default=1, help="1: the AR partitions are one CU range used by --ar-parts AR phases at once")
    ap.add_argument("--bulk-slots", type=int, default=1, help="refinement / decoding phases allowed at the same time on the throughput partition")
    ap.add_argument("--ar-parts", type=int, default=2, help="independent AR partitions (concurrent AR phases) when lanes > 1")
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per step (default: BASELINE configs[1])")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 (default): the parity configuration; bf16: bf16 weights/operands with fp32 accumulation (BASELINE configs[1] wording)")
    ap.add_argument("--frames", type=int, default=FRAMES, help="frames per utterance (default: BASELINE configs[1]; 400 = the long-form case)")
    ap.add_argument("--coalesce", default="auto", help="consecutive batches the pipeline generates / refines / decodes as ONE pass (per-utterance results are unchanged): "
                    "an integer, or 'auto' = sized by the queue depth (2 per pass for short queues, 4 from 4 queued batches per lane)")
    ap.add_argument("--voices", type=int, default=-1, help="distinct reference voices per batch (default: one per utterance, SURVEY 8d; 1 = one shared voice)")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs of the default run (one shared voice, 32x400 / 1x400 frames, bf16 mode)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--leg", type=str, default="", help=argparse.SUPPRESS)  # child process of the default run: one extra leg (see LEGS)
    ap.add_argument("--parity-tokens", type=str, default="", help=argparse.SUPPRESS)
    ap.add_argument("--input-rank", type=int, default=-1, help=argparse.SUPPRESS)  # tests: a 1-GPU run on the inputs of rank R
    args = ap.parse_args()
    BATCH, FRAMES = int(args.batch), int(args.frames)
    # (long-form jobs coalesce like short ones: the decoder takes a 64 x 400 pass in two 32-row chunks, sopro_mimi_chunk_rows)
    COALESCE = coalesce_arg(args.coalesce)

    if args.cpu_baseline_only:  # child process of the N=1 run: CPU only, bounded by the parent's timeout
        from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
        from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

        cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
        wn = synth_sopro_weights(cfg, VOCAB, 0, suppress_eos=True)
        par = oracle_parity(cfg, wn, args.parity_tokens) if args.parity_tokens else None
        base = cpu_baseline(cfg, mc, wn, synth_mimi_weights(mc, 0), args.cpu_utts)
        print(json.dumps({"cpu_baseline": base, "parity": par}), flush=True)
        return

    if args.gpus > 1 and "RANK" not in os.environ:  # started by hand: re-launch as one rank per GPU, like the driver does
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
                                   os.path.abspath(__file__), *sys.argv[1:]])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the Sopro engine has no CPU fallback)")
    # developer switch for a 1-GPU box: every rank on GPU 0, gloo for the barrier / MAX (exercises the launch and reduction path only)
    share_gpu = os.environ.get("SOPRO_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    from sopro_amd import hip

    # Host wait mode: a process-level decision, taken before the first stream exists (hip.set_host_wait).  The lanes' threads wait
    # most of the time (AR stop polls, phase ends): blocking waits instead of the runtime's spin free a core per lane - what 8
    # ranks on one host need.  SOPRO_BLOCKING_WAIT=0 keeps the spin wait (lowest wake-up latency for the TTFA leg).
    blocking_wait = (args.lanes > 1 or world > 1) and os.environ.get("SOPRO_BLOCKING_WAIT", "1") != "0"
    if blocking_wait:
        hip.set_host_wait(True, dev_index)
    if args.leg:  # child process of the default run: one extra leg, one JSON object on stdout
        print(json.dumps(leg_main(args.leg, args, device)), flush=True)
        return
    import torch.distributed as dist

    # One rank per GPU means 1 + lanes host threads per rank, all latency-sensitive (they keep the AR launch queues fed).  With
    # N ranks on one host every rank gets its own slice of the cores, so that no rank's launch threads are descheduled behind
    # another rank's; intra-op CPU threads are of no use to this path.  SOPRO_BENCH_PIN=0 leaves the placement to the OS.
    pinned = None
    if world > 1:
        torch.set_num_threads(1)
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if os.environ.get("SOPRO_BENCH_PIN", "1") != "0" and per >= 1:
            pinned = cores[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, pinned)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        # communicators are created lazily by the first collective: do that here, far away from the timed region
        warm_t = torch.zeros(1, device="cpu" if share_gpu else device)
        dist.all_reduce(warm_t, op=dist.ReduceOp.MAX)
        dist.barrier()

    from sopro_amd import hip

    log("building engine")
    tts, cfg, mc, wn, mn = build_engine(device, args.precision)
    in_rank = rank if args.input_rank < 0 else args.input_rank
    ids, ref_tq = make_inputs(in_rank)
    # per-voice preparation, outside the timed region (README "precalculate" flow); one voice per utterance
    n_voices = BATCH if args.voices < 0 else max(1, min(BATCH, args.voices))
    voices = [tts.prepare_reference(ref_tokens_tq=v) for v in make_voices(in_rank, n_voices)]
    ref = voices[0]
    refs = [voices[i % n_voices] for i in range(BATCH)]
    # one seed for every step: the steps are then the same job, and their outputs must be bit-identical (checked below)
    kw = dict(max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, text_ids=ids, seed=20260924)

    job = dict(texts=[""] * BATCH, refs=refs, **kw)
    kept = []  # device waveforms of the first and the last timed step (references only: nothing is copied in the timed region)

    def check(out):
        assert all(o.shape[-1] == FRAMES * 1920 for o in out)
        if len(kept) < 2:
            kept.append(out)
        else:
            kept[1] = out

    pipe = None
    if args.lanes > 1:
        from sopro_amd.pipeline import PipelinedSynthesizer

        try:
            pipe = PipelinedSynthesizer(tts, lanes=args.lanes, ar_cus=args.ar_cus, ar_parts=args.ar_parts, ar_shared=bool(args.ar_shared), bulk_slots=args.bulk_slots)
        except Exception as e:  # noqa: BLE001  (e.g. a device without CU-mask support): fall back to sequential batches
            log(f"pipelining unavailable ({e!r}); running sequential batches")
            pipe, args.lanes = None, 1

    def run_steps(n, timings=None):
        """n passes of the hot path over n independent batches (pipelined across lanes when lanes > 1)."""
        if pipe is None:
            for _ in range(n):
                check(tts.synthesize_batch(timings=timings, **job))
        else:
            for out in pipe.run([job] * n, timings=timings, coalesce=COALESCE):
                check(out)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("warmup")
    # every lane runs a shape eagerly once (scratch allocation) and records its launch sequences on the second pass
    # (a pipeline's warm-up reaches its steady state: EVERY lane has run a pass of the timed shape twice - eagerly, then recording its
    # launch sequences - i.e. 2 x lanes passes of `coalesce` jobs each; the driver's W is a lower bound and `warmup_run` says what ran.
    # SOPRO_BENCH_WARM_PASSES=1: the round-4 rule, one pass per lane, which left every lane's recording inside the timed region)
    sizes = sorted(set(pipe.pass_sizes(args.steps, COALESCE))) if pipe is not None else [1]
    if pipe is not None:
        pipe.prepare(job, sizes=sizes)  # deterministic: every lane, every pass size of the timed run, twice (eager, then recording)
    warm = max(args.warmup, 2)
    run_steps(warm)
    warm_desc = (f"every lane ran a pass of {' / '.join(str(v) for v in sizes)} batch(es) twice (PipelinedSynthesizer.prepare), then {warm} pipelined steps"
                 if pipe is not None else f"{warm} sequential steps")
    fence()
    kept.clear()
    log("timed steps")
    hip.phase_log = []  # two HIP events per AR phase, on the AR stream: the frame time of the timed region itself
    phases = {}
    c0 = time.process_time()  # CPU time of every thread of this process (launch threads included)
    t0 = time.perf_counter()
    run_steps(args.steps, phases)
    fence()
    dt = time.perf_counter() - t0
    host_cpu = time.process_time() - c0
    log(f"timed region done: {dt:.3f} s for {args.steps} steps; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if os.environ.get("SOPRO_BENCH_MEMCENSUS", "0") == "1":  # developer: who holds the device memory (stderr)
        lanes_ = pipe.lanes if pipe is not None else [tts]
        for li, ln in enumerate(lanes_):
            for nm, ws in (("model", ln.model.ws), ("codec", ln.codec.ws)):
                top = sorted(((t.numel() * t.element_size(), k[0], k[1]) for k, t in ws._bufs.items()), reverse=True)[:6]
                log(f"lane {li} {nm}.ws {ws.bytes / 2**30:.2f} GiB: " + "; ".join(f"{n} {list(sh)} {b / 2**30:.2f}" for b, n, sh in top))
        log(f"allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if os.environ.get("SOPRO_BENCH_TRACE") and pipe is not None:  # developer aid: which lane ran which step when, and its phase times
        for rep in range(int(os.environ["SOPRO_BENCH_TRACE"])):
            if rep:
                ph2 = {}
                t1 = time.perf_counter()
                pipe.run([job] * args.steps, timings=ph2)
                fence()
                log(f"trace repeat {rep}: {time.perf_counter() - t1:.3f} s")
            for i, lane, a, b, tj in sorted(pipe.trace):
                log(f"  step {i:3d} lane {lane} {a * 1e3:8.1f} -> {b * 1e3:8.1f} ms  " + " ".join(f"{k}={v * 1e3:.1f}" for k, v in tj.items() if not k.startswith("_")))
            bl = sorted((tj["_bulk_t0"], tj["_bulk_t1"]) for *_x, tj in pipe.trace if "_bulk_t0" in tj)
            gaps = [(b0 - a1) * 1e3 for (_a0, a1), (b0, _b1) in zip(bl, bl[1:])]
            if gaps:
                log("  refinement + decode slot idle between batches (host clock, ms): " + " ".join(f"{g:.2f}" for g in gaps))
            for part in range(args.ar_parts):  # lane i generates in slot i % ar_parts
                al = sorted((tj["_ar_t0"], tj["_ar_t1"]) for _i, lane, _a, _b, tj in pipe.trace if "_ar_t0" in tj and lane % args.ar_parts == part)
                g2 = [(b0 - a1) * 1e3 for (_a0, a1), (b0, _b1) in zip(al, al[1:])]
                log(f"  generation slot {part} idle between phases (host clock, ms): " + " ".join(f"{g:.2f}" for g in g2))
    ar_log, hip.phase_log = hip.phase_log, None
    ar_frames = sum(n for n, _b, _e0, _e1 in ar_log)
    ar_ms = sum(e0.elapsed_time(e1) for _n, _b, e0, e1 in ar_log)
    steps_identical = bool(len(kept) == 2 and all(torch.equal(a, b) for a, b in zip(kept[0], kept[1])))
    finite = bool(kept and all(bool(torch.isfinite(o).all()) for o in kept[-1]))
    # content hash of this rank's last step (PCM16 of every utterance): N ranks must reproduce what N single-GPU runs give
    import hashlib

    from sopro_amd.wire import float_to_pcm16le

    hsh = hashlib.sha256()
    for o in (kept[-1] if kept else []):
        hsh.update(float_to_pcm16le(o.reshape(1, -1)))
    rank_hashes = [hsh.hexdigest()[:16]]
    if world > 1:
        allh = [None] * world
        dist.all_gather_object(allh, rank_hashes[0])
        rank_hashes = allh
    kept.clear()
    host_cpu_ranks = [round(host_cpu / args.steps, 4)]
    if world > 1:
        allc = [None] * world
        dist.all_gather_object(allc, host_cpu_ranks[0])
        host_cpu_ranks = allc
        t = torch.tensor([dt], device="cpu" if share_gpu else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- instrumented repeat of the same steps (same lanes, streams and CU partitions): HIP events around every GEMM /
    # attention launch and AR frame replay, recorded on the stream of the launch.  The recorded NAR / Mimi launch sequences
    # are issued eagerly here so that their launches are visible one by one; the timed region above is not instrumented.
    prof = hip.Profiler()
    phases_prof: dict = {}  # phase times (device events) of the instrumented repeat itself
    nprof = max(0, min(args.profile_steps, args.steps))
    if nprof > 0:
        hip.set_profiler(prof)
        run_steps(nprof, phases_prof)
        fence()
        hip.set_profiler(None)

    # ---- rooflines of the instrumented families (HIP events recorded on the engine streams during the instrumented repeat)
    fam = prof.summary()
    # The instrumented repeat issues the recorded refinement / decoder sequences EAGERLY, with two events around every heavy launch:
    # its launches are slower than the same launches replayed from the recorded sequence (round 5: a family's instrumented time
    # exceeded the timed region's whole phase).  What the repeat measures reliably is each family's SHARE of its phase; the time
    # that counts is the phase's own device-event time in the timed region.  So: family time = instrumented family time x
    # (timed phase per step / instrumented phase per step) - by construction a family never exceeds its phase.
    fam_phase = {"gemm_f16x3_kernel": "nar", "gemm_bf16x3_kernel": "mimi", "seanet_uptail_kernel": "mimi", "seanet_tail_kernel": "mimi",
                 "seanet_res128_kernel": "mimi", "seanet_up128_kernel": "mimi", "attention_split_kernel": "mimi",
                 "gemm_f32_kernel": "cond", "gemm_bf16x6_kernel": "cond", "attention_kernel": "cond"}
    bulk = ("cond", "nar", "mimi")
    scale_of: dict = {}
    if nprof > 0:
        for ph in bulk:
            if phases.get(ph, 0.0) > 0 and phases_prof.get(ph, 0.0) > 0:
                scale_of[ph] = (phases[ph] / args.steps) / (phases_prof[ph] / nprof)
        tot_t, tot_p = sum(phases.get(k, 0.0) for k in bulk) / args.steps, sum(phases_prof.get(k, 0.0) for k in bulk) / nprof
        scale_of[None] = (tot_t / tot_p) if tot_t > 0 and tot_p > 0 else 1.0
    rocf = rocprof_family_table()
    for k, v in fam.items():
        if k == "ar_step_graph":
            continue
        ph = fam_phase.get(k)
        sc = scale_of.get(ph, scale_of.get(None, 1.0))
        v["ms_instrumented"], v["phase"], v["phase_scale"] = v["ms"], ph, round(sc, 4)
        if 0.4 <= sc <= 1.6:
            v["ms"] = v["ms"] * sc
        elif k in rocf.get("families", {}):
            # the instrumented repeat's phase is not the timed region's (seen on a busy host: a conditioning phase that waited for its
            # turn 20 x longer inside the repeat, which scaled its attention launch to 6 us and 3.8 x the matrix cores' peak): the family's
            # time is then the committed rocprof average of its launches, and the entry says so
            v["ms"] = rocf["families"][k] * 1e-3 * max(1, v["launches"])
            v["phase_scale"], v["timing"] = None, "rocprof average (instrumented phase / timed phase = %.3g: unrepresentative repeat)" % sc
    pmc_d, busy_d, ark = latest_profile("pmc_summary.json"), latest_profile("pmc_mfma_busy.json"), latest_profile("ar_kernels.json")
    pmc, busy = pmc_d.get("families", {}), busy_d.get("families", {})
    stamp_of = lambda d: profile_stamp(os.path.join(ROOT, d["source"])) if d.get("source") else None  # noqa: E731
    share = (256 - args.ar_cus * (1 if args.ar_shared else args.ar_parts)) / 256.0 if (args.lanes > 1 and args.ar_cus > 0) else 1.0  # CUs of the bulk partition
    ar_share = (args.ar_cus * (1 if args.ar_shared else args.ar_parts)) / 256.0 if (args.lanes > 1 and args.ar_cus > 0) else 1.0
    measured = (f"HIP events on the launch stream over an instrumented repeat of {nprof} steps right after the timed region (host-bound samples "
                "excluded), scaled by timed-phase / instrumented-phase device time so that a family is a share of its phase in the timed region")

    def mfma_entry(key, title, peak, passes, extra_note):
        f = fam[key]
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
        e = {"kernel": title, "bound": "mfma", "achieved": round(ach, 3), "peak": round(peak * share, 2), "unit": "TFLOP/s",
             "frac": round(ach / (peak * share), 5), "cu_share": share, "peak_full_chip": peak, "frac_full_chip": round(ach / peak, 5),
             "traffic": pmc.get(key, {}).get("traffic_bytes_per_launch"), "launches": f["launches"],
             "avg_launch_us": round(f["ms"] / max(1, f["launches"]) * 1e3, 2), "ms_per_step": round(f["ms"] / max(1, nprof), 3),
             "algorithmic_flops_per_launch": round(f["flops"] / max(1, f["launches"])), "measured": measured,
             "phase": f.get("phase"), "phase_scale": f.get("phase_scale"), **({"timing": f["timing"]} if f.get("timing") else {}),
             "avg_launch_us_instrumented": round(f.get("ms_instrumented", f["ms"]) / max(1, f["launches"]) * 1e3, 2),
             "gpu_bound_samples": f.get("gpu_bound"), "samples": f["launches"]}
        if passes > 1:
            e["mfma_passes_per_product"] = passes
            e["frac_of_pass_ceiling"] = round(ach / (peak * share / passes), 5)
            e["note"] = f"achieved = algorithmic (fp32-equivalent) flops / launch time; each product costs {passes} bf16 MFMA passes. " + extra_note
        if e["traffic"] is not None:
            e["traffic_profile"] = stamp_of(pmc_d)
        if key in busy:
            e["mfma_busy_pmc"] = busy[key].get("mfma_busy_share_at_2p4ghz")
            e["mfma_busy_profile"] = stamp_of(busy_d)
        if key in rocf.get("families", {}):
            e["avg_launch_us_rocprof"] = rocf["families"][key]
            e["rocprof_profile"] = rocf["stamp"]
        return e

    entries = []  # (summed ms, entry)
    if "gemm_bf16x3_kernel" in fam:
        entries.append((fam["gemm_bf16x3_kernel"]["ms"], mfma_entry(
            "gemm_bf16x3_kernel", "gemm_bf16x3_kernel (Mimi decoder contractions; v_mfma_f32_32x32x16_bf16, 3 passes per product)",
            PEAK_BF16_MFMA_TFLOPS, 3, "16-bit operands, waveform contract 1e-4 of peak")))
    if "gemm_bf16x6_kernel" in fam:
        entries.append((fam["gemm_bf16x6_kernel"]["ms"], mfma_entry(
            "gemm_bf16x6_kernel", "gemm_bf16x6_kernel (NAR contractions; v_mfma_f32_32x32x16_bf16, 6 passes per product, 24-bit operands)",
            PEAK_BF16_MFMA_TFLOPS, 6, "fp32-class products: refined tokens must equal the fp32 reference's")))
    if "gemm_f16x3_kernel" in fam:
        entries.append((fam["gemm_f16x3_kernel"]["ms"], mfma_entry(
            "gemm_f16x3_kernel", "gemm_f16x3_kernel (NAR contractions; v_mfma_f32_32x32x16_f16, 3 passes per product, two fp16 pieces = 22-bit operands)",
            PEAK_BF16_MFMA_TFLOPS, 3, "fp32-class products (as accurate as the six-pass bf16 form: profiles/r03_f16x3_probe.txt): refined tokens must equal the fp32 reference's")))
    if "gemm_bf16x1_kernel" in fam:
        entries.append((fam["gemm_bf16x1_kernel"]["ms"], mfma_entry(
            "gemm_bf16x1_kernel", "gemm_bf16x1_kernel (bf16 mode: NAR + Mimi contractions, bf16 operands, fp32 accumulate, one MFMA pass)",
            PEAK_BF16_MFMA_TFLOPS, 1, "")))
    if "gemm_f32_kernel" in fam:
        entries.append((fam["gemm_f32_kernel"]["ms"], mfma_entry(
            "gemm_f32_kernel", "gemm_f32_kernel (all tile shapes; v_mfma_f32_32x32x2_f32)", PEAK_F32_MFMA_TFLOPS, 1, "")))
    mimi_passes = 1 if args.precision == "bf16" else 3
    for key, what in (("seanet_uptail_kernel", "the whole last level in one kernel: last transposed convolution + last residual block + last layer"),
                      ("seanet_tail_kernel", "fused 24 kHz tail: last residual block + final convolution"),
                      ("seanet_res128_kernel", "fused residual block of the 128-channel level"),
                      ("seanet_up128_kernel", "weight-stationary last transposed convolution")):
        if key in fam:
            entries.append((fam[key]["ms"], mfma_entry(key, f"{key} ({what}; v_mfma_f32_32x32x16_bf16, {mimi_passes} pass(es) per product)",
                                                       PEAK_BF16_MFMA_TFLOPS, mimi_passes, "16-bit operands, waveform contract 1e-4 of peak")))
    if "attention_kernel" in fam and fam["attention_kernel"]["flops"] > 0:
        entries.append((fam["attention_kernel"]["ms"], mfma_entry(
            "attention_kernel", "attention_kernel family (attn_mfma_kernel: reference cross-attention of the conditioning, exact fp32; v_mfma_f32_32x32x2_f32)",
            PEAK_F32_MFMA_TFLOPS, 1, "")))
    if "attention_split_kernel" in fam and fam["attention_split_kernel"]["flops"] > 0:
        entries.append((fam["attention_split_kernel"]["ms"], mfma_entry(
            "attention_split_kernel", "attn_mfma_split_kernel (codec decoder window attention, two bf16 pieces per operand; v_mfma_f32_32x32x16_bf16, 3 passes per product)",
            PEAK_BF16_MFMA_TFLOPS, 3, "16-bit operands, waveform contract 1e-4 of peak")))
    if "ar_step_graph" in fam:
        f = fam["ar_step_graph"]
        inst_ms = f["ms"] / max(1, f["launches"])  # instrumented repeat (events around every replay; NAR / Mimi issued eagerly)
        per_launch_ms = ar_ms / ar_frames if ar_frames else inst_ms  # the timed region itself: phase events / frames
        # rows of a frame: the passes of a run may differ in size (pass_sizes: a single-batch first pass, then coalesced ones) - the
        # achieved rate is the algorithmic bytes of EVERY replayed frame at its own row count over the summed phase times; the
        # per-launch figures describe the most common frame
        wb = 2 if args.precision == "bf16" else 4
        by_rows = {}
        for n_, b_, _e0, _e1 in ar_log:
            by_rows[b_] = by_rows.get(b_, 0) + n_
        rows_launch = max(by_rows, key=by_rows.get) if by_rows else BATCH
        bytes_step = ar_step_bytes(rows_launch, TEXT_LEN, wb)
        bytes_all = sum(n_ * ar_step_bytes(b_, TEXT_LEN, wb) for b_, n_ in by_rows.items()) if by_rows else bytes_step * max(1, f["launches"])
        ach = (bytes_all / (ar_ms * 1e-3) / 1e9) if ar_frames else bytes_step / (per_launch_ms * 1e-3) / 1e9
        per_frame = ark.get("launches_per_frame") or {"skinny_kernel": 19, "xattn_step_kernel": 3, "ar_sample_kernel": 1}
        # round 5: the counters were also collected on the PIPELINED run (tools/collect_evidence.sh: rNN_pmc_summary_pipelined.json, four
        # lanes, CU partitions, coalesced passes) - when its frames have this run's row count, the traffic figures are THAT pass's
        pmc_p = latest_profile("pmc_summary_pipelined.json")
        if pmc_p.get("families") and int(pmc_p.get("ar_rows_per_frame", 0)) == rows_launch and pmc_p.get("precision", "f32") == args.precision:
            pmc_d, pmc = pmc_p, pmc_p["families"]
        tr = round(sum(pmc.get(k, {}).get("traffic_bytes_per_launch", 0) * n for k, n in per_frame.items())) if pmc else None
        # the PMC pass has its own row count per frame (profiles/rNN_pmc_summary.json "ar_rows_per_frame"; the r03 file was a
        # 32-row --lanes 1 run): traffic is compared with the algorithmic bytes of THAT frame, not of this run's coalesced one
        pmc_rows = int(pmc_d.get("ar_rows_per_frame", 32)) if pmc else None
        pmc_prec = pmc_d.get("precision", "f32") if pmc else None
        tr_alg = ar_step_bytes(pmc_rows, TEXT_LEN, 2 if pmc_prec == "bf16" else 4) if pmc else None
        e = {"kernel": f"AR frame (hipGraph of {sum(per_frame.values())} launches: " + ", ".join(f"{k} x{n}" for k, n in per_frame.items()) + ")",
             "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 5),
             "traffic": tr or None, "traffic_ratio": round(tr / tr_alg, 3) if tr else None,
             "traffic_rows_per_launch": pmc_rows if tr else None, "traffic_algorithmic_bytes": tr_alg if tr else None,
             "traffic_note": (f"PMC pass of a {pmc_rows}-row {pmc_prec} frame ({pmc_d.get('source')}; dispatches are serialised by the counter collection, "
                              f"so the two partitions do not overlap under it): traffic and traffic_ratio describe that frame; achieved / avg_launch_us "
                              f"describe this run's {rows_launch}-row frames") if tr else None,
             "launches": ar_frames or f["launches"],
             "avg_launch_us": round(per_launch_ms * 1e3, 2), "ms_per_step": round((ar_ms / args.steps) if ar_frames else f["ms"] / max(1, nprof), 3),
             "avg_launch_us_instrumented_repeat": round(inst_ms * 1e3, 2),
             "algorithmic_bytes_per_launch": bytes_step, "rows_per_launch": rows_launch, "frames_by_rows": {str(k): v for k, v in sorted(by_rows.items())},
             "cu_share": ar_share, "concurrent_phases": (args.ar_parts if args.lanes > 1 else 1),
             "wall_ms_per_step": round(dt / args.steps * 1e3, 3),
             "frac_reading": "a chain of 23 dependent launches per frame (latency-bound, 3-6 % of HBM peak by construction; batch1 sits on the "
                             "launch-per-stage floor); with 4 jobs per pass this partition has slack: throughput is bound by roofline_more",
             "measured": ("two HIP events per AR phase on the AR stream IN the timed region: sum of phase times / frames replayed"
                          if ar_frames else measured),
             "note": (f"one launch = one frame of one {rows_launch}-row pass; in the pipeline two AR phases replay concurrently on the generation "
                      "partition, so the phase sum exceeds the wall time per step" if args.lanes > 1 else "sequential batches, whole chip")}
        if tr:
            e["traffic_profile"] = stamp_of(pmc_d)
        if ark.get("kernels"):
            e["per_kernel_us_rocprof"] = ark["kernels"]
            e["per_kernel_profile"] = profile_stamp(os.path.join(ROOT, ark["source"])) if ark.get("source") and not os.path.isabs(ark["source"]) \
                else {"file": ark.get("source"), "replayed": True}
        entries.append(((ar_ms * nprof / args.steps) if ar_frames else f["ms"], e))  # same basis as the others: ms over nprof steps
    # a contraction family cannot have spent more than the phases that contain it (conditioning + refinement + decoding, from the
    # un-instrumented phase timers of the timed region): caps what a host-bound instrumented repeat could inflate
    bulk_cap = sum(phases.get(k, 0.0) for k in ("cond", "nar", "mimi")) / max(1, args.steps) * 1e3 * max(1, nprof)
    entries = [((min(ms, bulk_cap) if (bulk_cap > 0 and not e["kernel"].startswith("AR frame")) else ms), e) for ms, e in entries]
    entries.sort(key=lambda t: -t[0])
    # Families whose launches are short (rocprof average under 50 us) are NOT reported from HIP events: events around a short
    # launch on a shared partition mostly measure queueing (r2: 187 us per launch by events, 32.8 us by rocprof).  They keep
    # their launch count and the rocprof figure of the committed profile.
    small = {k for k, us in rocf.get("families", {}).items() if us < 50.0}
    dropped = [e["kernel"] for _, e in entries if any(e["kernel"].startswith(k) for k in small)]
    entries = [(ms, e) for ms, e in entries if not any(e["kernel"].startswith(k) for k in small)]
    roof = entries[0][1] if entries else None
    roof_more = [e for _, e in entries[1:]]
    families = {}
    for k, v in fam.items():
        d = {"launches_per_step": v["launches"] // max(1, nprof)}
        if k in small:
            d.update(timing="rocprof (committed profile): launches under 50 us are not timed with HIP events", avg_launch_us_rocprof=rocf["families"][k],
                     ms_per_step_rocprof=round(rocf["families"][k] * d["launches_per_step"] / 1e3, 3), rocprof_profile=rocf["stamp"])
        else:
            d.update(timing="hip-events (instrumented repeat)", ms_per_step=round(v["ms"] / max(1, nprof), 3),
                     tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 3) if v["flops"] else None)
            if k in rocf.get("families", {}):
                d["avg_launch_us_rocprof"] = rocf["families"][k]
        families[k] = d

    pipe_sizes = pipe.pass_sizes(args.steps, COALESCE) if pipe is not None else None
    # passes of this process (warm-up, timed, instrumented) whose f16 refinement operands left fp16's range and were repeated on
    # the six-pass operands (sopro_gemm_split_ext.range_events, SoproTTSModel.nar_guard): 0 on an in-range checkpoint
    range_fallbacks = sum(getattr(l.model, "range_fallbacks", 0) for l in (pipe.lanes if pipe is not None else [tts]))
    if pipe is not None:
        pipe.close()  # the latency leg below runs on the whole chip

    # ---- the AR frame alone on the whole chip (nothing else running): what the frame costs without the pipeline's contention
    iso = None
    if rank == 0 and roof is not None and "ar_step_graph" in fam:
        log("isolated AR frame")
        prep = tts.model.phase_cond(ids, refs, max_frames=FRAMES - 1, style_strength=float(cfg.style_strength))
        arkw = dict(max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, style_strength=float(cfg.style_strength),
                    min_gen_frames=None, prep=prep, seed=1)
        tts.model.phase_ar(ids, refs, **arkw)  # records the frame graph of this stream
        p2 = hip.Profiler()
        hip.set_profiler(p2)
        tts.model.phase_ar(ids, refs, **arkw)
        torch.cuda.synchronize()
        hip.set_profiler(None)
        f = p2.summary().get("ar_step_graph")
        if f:
            us = f["ms"] / max(1, f["launches"]) * 1e3
            bts = ar_step_bytes(BATCH, TEXT_LEN, 2 if args.precision == "bf16" else 4)
            iso = {"avg_launch_us": round(us, 2), "achieved_GBps": round(bts / (us * 1e-6) / 1e9, 1),
                   "frac": round(bts / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 5), "launches": f["launches"],
                   "what": "one 32-row AR phase alone on all 256 CUs, HIP events around every frame-graph replay"}
        for e in [roof] + roof_more:
            if e["kernel"].startswith("AR frame"):
                e["isolated_whole_chip"] = iso

    # ---- one greedy batch of the same shape for the parity leg (row 0 goes to the oracle in the CPU child below)
    parity_path = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import tempfile

        log("greedy parity batch")
        gt = tts.model.generate_tokens_batch(ids, refs, max_frames=FRAMES - 1, top_p=0.0, temperature=1.0, anti_loop=False,
                                             style_strength=float(cfg.style_strength))
        parity_path = os.path.join(tempfile.mkdtemp(prefix="sopro_bench_"), "row0.npy")
        np.save(parity_path, gt[0].cpu().numpy())

    # ---- p50 time-to-first-audio of stream(), batch 1 (BASELINE.json configs[2]); outside the timed steps
    ttfa = None
    if rank == 0 and args.ttfa_runs > 0:
        log("ttfa")
        lat = []
        TTFA_WARM = 5
        for i in range(args.ttfa_runs + TTFA_WARM):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            it = tts.stream("", ref=ref, max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, chunk_frames=6,
                            text_ids=ids[i % BATCH])
            first = next(it)
            torch.cuda.synchronize()
            if i >= TTFA_WARM:
                lat.append((time.perf_counter() - t1) * 1e3)
            it.close()
            del it, first
        ttfa = float(np.percentile(lat, 50))

    # ---- extra legs of the default run (BASELINE.json configs the headline does not cover), rank 0 of a single-GPU run only.
    # Every leg runs in a CHILD PROCESS of its own (`bench.py --leg NAME`): a leg that dies - call 7 of round 4: SIGSEGV inside
    # torch's caching allocator in the fifth leg of one process, whose earlier legs had left cached blocks of destroyed CU-masked
    # streams behind - must not take the headline with it, and every leg starts from an empty device.  The parent's engines are
    # released first.
    legs = None
    default_shape = (BATCH, FRAMES) == (32, 200) and args.precision == "f32"
    if rank == 0 and world == 1 and default_shape and not args.no_legs:
        import subprocess

        legs = {}
        job.clear()
        kept.clear()
        del tts, voices, refs, ref
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        for name in LEGS:
            log(f"leg: {name} (child process)")
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--steps", str(args.steps), "--lanes", str(args.lanes),
                                    "--ar-cus", str(args.ar_cus), "--ar-parts", str(args.ar_parts), "--ar-shared", str(args.ar_shared),
                                    "--bulk-slots", str(args.bulk_slots), "--coalesce", str(args.coalesce)],
                                   capture_output=True, text=True, timeout=240)
                legs[name] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001  (a failed leg must not void the headline)
                log(f"leg {name} failed: {e!r}")
                legs[name] = {"error": repr(e)}

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess

        log("cpu baseline + oracle parity (child process, <= 180 s)")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-utts", str(args.cpu_utts), "--frames", str(FRAMES),
                                "--batch", str(BATCH), "--parity-tokens", parity_path or ""],
                               capture_output=True, text=True, timeout=180, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            both = json.loads(r.stdout.strip().splitlines()[-1])
            cpu, parity = both["cpu_baseline"], both["parity"]
            rr = latest_profile("cpu_reference_ratio.json")  # oracle vs THE REFERENCE on one host (build container: tools/cpu_reference_ratio.py)
            if cpu and rr.get("oracle_over_reference"):
                cpu["reference_ratio"] = {"oracle_over_reference": rr["oracle_over_reference"], "reference_audio_s_per_s": rr.get("reference_audio_s_per_s"),
                                          "oracle_audio_s_per_s": rr.get("oracle_audio_s_per_s"), "host": rr.get("host"),
                                          "note": "kind = 'port': /root/reference is not on the GPU box; on the build container the port runs this much "
                                                  "faster than the reference itself on the same utterances, so the reference on THIS host would be about "
                                                  "value / oracle_over_reference",
                                          "estimated_reference_value_here": round(cpu["value"] / rr["oracle_over_reference"], 3),
                                          **profile_stamp(os.path.join(ROOT, rr["source"]))}
        except Exception as e:  # noqa: BLE001  (a missing baseline must not void the GPU measurement)
            log(f"cpu baseline failed: {e!r}")
            cpu = None
    parity = dict(parity or {}, timed_steps_identical=steps_identical, timed_outputs_finite=finite, rank_output_sha16=rank_hashes,
                  f16_range_fallbacks=range_fallbacks,
                  how="all timed steps run one seeded job: first vs last step compared bit for bit; oracle leg: row 0 of a greedy batch of "
                      "the same shape vs oracle/sopro_oracle.py (codebook 0 exact; refined tokens exact or audited near-ties < 1e-4)")
    if args.precision != "f32":  # the bf16 mode is a throughput mode with its own quality numbers (tests/test_gpu_bf16_mode.py), not a parity mode
        parity["mode"] = f"{args.precision}: not a parity mode - 'ok' compares with the fp32 oracle and is expected to be false"

    if rank == 0:
        audio_sec = world * args.steps * BATCH * FRAMES * FRAME_SEC
        # what the driver's record keeps are the `config`, `roofline` and `cpu_baseline` objects: BASELINE's second metric (p50 TTFA)
        # and the other configs' values ride in them as compact summaries (the full leg objects stay under "legs")
        legs_summary = None
        if legs:
            legs_summary = {k: ({"value": v.get("value"), "unit": v.get("unit"), "ms_per_step": v.get("ms_per_step"), "dtype": v.get("dtype", "f32"),
                                 "batch": v.get("batch"), "frames": v.get("frames"), "lanes": v.get("lanes")} if "value" in v else {"error": v.get("error")})
                            for k, v in legs.items()}
            b1 = (legs.get("f32_1x400_sequential") or {}).get("roofline")
            if b1 and roof is not None:
                roof["batch1"] = b1
        second = {"ttfa_ms_p50": None if ttfa is None else round(ttfa, 3), "cpu_ttfa_ms_p50": (cpu or {}).get("ttfa_ms_p50"),
                  "what": "p50 time to first audio of stream(), batch 1, chunk_frames 6, hipGraph-replayed AR frames (BASELINE configs[2]); "
                          "cpu = the oracle port on this host's cores"}
        line = {
            "metric": "audio_seconds_per_second", "value": round(audio_sec / dt, 2), "unit": "audio-s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "warmup_run": warm, "warmup_detail": warm_desc, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"Sopro-135M synthesize, {BATCH} utterances x {FRAMES} frames per GPU ({'BASELINE configs[1]' if (BATCH, FRAMES) == (32, 200) else 'non-default shape'}), "
                                   f"S={TEXT_LEN} text tokens, {n_voices} distinct {REF_FRAMES}-frame reference voice(s) per batch prepared outside the timed "
                                   "region, top_p=0.9 T=1.05 anti_loop (reference defaults), synthetic weights with EOS suppressed",
                       "batch_per_gpu": BATCH, "frames": FRAMES, "voices_per_batch": n_voices,
                       "parallelism": f"replicas x{world} (utterance sharding, no collective)",
                       "lanes_per_gpu": args.lanes, "coalesce": (COALESCE if args.lanes > 1 else 1),
                       "pass_sizes": (pipe_sizes if args.lanes > 1 else None),
                       "second_metric": second, "legs": legs_summary,
                       "pipelining": (f"{args.lanes} engines per GPU share the weights: up to {args.ar_parts} AR phases at a time on "
                                      f"{'one shared partition' if args.ar_shared else 'partitions'} of {args.ar_cus} CUs while conditioning, NAR and Mimi "
                                      f"decode of other batches run on the other {int(round(256 * share))} CUs (hipExtStreamCreateWithCUMask); NAR and Mimi "
                                      "launch sequences are recorded hipGraphs"
                                      + (f"; consecutive batches are coalesced into passes of {pipe_sizes} batches (every utterance keeps its "
                                         "own sampler stream: outputs are bit-identical to un-coalesced steps)" if pipe_sizes and max(pipe_sizes) > 1 else "")) if args.lanes > 1 else "none"},
            "dtype_detail": ("fp32 tensors and accumulation everywhere; conditioning + AR on v_mfma_f32_*_f32 (round 4: the text cross-attention reads unfolded "
                             "keys, its query rides on the feed-forward launches; the two encoders' contractions on three "
                             "bf16 pieces / 6 MFMA passes, 24 mantissa bits); NAR contractions with operands split into two fp16 pieces (22 mantissa bits, "
                             "3 MFMA passes, power-of-two operand scaling: as accurate as the six-pass bf16 form, profiles/r03_f16x3_probe.txt); Mimi decoder "
                             "contractions with two bf16 pieces (16 bits, 3 passes)")
                            if args.precision == "f32" else
                            ("bf16 mode (SURVEY 8d config 2; round 4: bf16 in memory): NAR + Mimi contractions one MFMA pass on bf16 operands, the SEANet decoder's "
                             "activations as bf16 rows in memory, the AR frame with bf16 weights, bf16 folded text operands and bf16 ring buffers; fp32 accumulators, "
                             "norms, softmax, residual streams, codec transformer stream; conditioning stays fp32.  Not a parity line: see tests/test_gpu_bf16_mode.py"),
            "phase_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phases.items()},
            "kernel_families": families,
            "ttfa_ms_p50": None if ttfa is None else round(ttfa, 3),
            "cpu_ttfa_ms_p50": (cpu or {}).get("ttfa_ms_p50"),
            "host_cpu_s_per_step": round(host_cpu / args.steps, 4),
            "host_cpu_s_per_step_by_rank": host_cpu_ranks,
            "host_threads": {"per_rank": 1 + (args.lanes if args.lanes > 1 else 0), "torch_intra_op": torch.get_num_threads(),
                             "pinned_cores_rank0": (len(pinned) if pinned else None)},
            "host_cpu_note": "CPU time of all threads of this rank's process over the timed region / steps (launch threads of the lanes included)",
            "host_wait": "blocking (hipDeviceScheduleBlockingSync, set at process start)" if blocking_wait else "spin (runtime default)",
            "legs": legs, "roofline_dropped": dropped,
            "unpinned": ["sinc_resample of the reference-audio front end (SURVEY 8f rank 1, outside this line's timed path): pinned by derivation "
                         "only - torchaudio is not in the image to generate a fixture from"],
            "roofline": roof, "roofline_more": roof_more, "cpu_baseline": cpu, "parity": parity,
        }
        emit(line, ROOT)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
