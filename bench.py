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
  roofline     -- the dominant kernel family of the step (by summed HIP-event time on the engine streams),
                  its algorithmic flops (MFMA bound) or bytes (HBM bound) per launch / its average launch time
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


def build_engine(device: str):
    from sopro_amd import SoproTTS
    from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
    from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    wn = synth_sopro_weights(cfg, VOCAB, 0, suppress_eos=True)  # EOS bias -1e9: fixed-length runs
    mn = synth_mimi_weights(mc, 0)
    return SoproTTS.from_weights(cfg, wn, mn, Tok(), device=device), cfg, mc, wn, mn


def make_inputs(rank: int):
    rng = np.random.default_rng(1000 + rank)
    ids = [torch.from_numpy(rng.integers(0, VOCAB, size=TEXT_LEN)) for _ in range(BATCH)]
    ref_tq = torch.from_numpy(rng.integers(0, 2048, size=(REF_FRAMES, 32)))
    return ids, ref_tq


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
    return {"value": round(frames * FRAME_SEC / dt, 3), "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_utts} utterances x {FRAMES} frames, sequential synthesize() of oracle/sopro_oracle.py (torch fp32 CPU), "
                      f"{dt:.1f} s wall"}


def ar_step_bytes(B: int, S: int) -> float:
    """weights + cond/embedding rows + ring buffers + K/V, fp32 (w = a = 4 bytes)."""
    return 10_575_492 * 4 + B * (768 + 32_256 + 2304 * S) * 4


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
    ap.add_argument("--ttfa-runs", type=int, default=20)
    ap.add_argument("--lanes", type=int, default=4, help="engines pipelined on one GPU (1 = strictly sequential batches)")
    ap.add_argument("--ar-cus", type=int, default=64, help="CUs of each AR partition (latency-bound phase) when lanes > 1")
    ap.add_argument("--ar-shared", type=int, default=1, help="1: the AR partitions are one CU range used by --ar-parts AR phases at once")
    ap.add_argument("--bulk-slots", type=int, default=1, help="refinement / decoding phases allowed at the same time on the throughput partition")
    ap.add_argument("--ar-parts", type=int, default=2, help="independent AR partitions (concurrent AR phases) when lanes > 1")
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per step (default: BASELINE configs[1])")
    ap.add_argument("--frames", type=int, default=FRAMES, help="frames per utterance (default: BASELINE configs[1]; 400 = the long-form case)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    BATCH, FRAMES = int(args.batch), int(args.frames)

    if args.cpu_baseline_only:  # child process of the N=1 run: CPU only, bounded by the parent's timeout
        from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
        from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

        cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
        print(json.dumps(cpu_baseline(cfg, mc, synth_sopro_weights(cfg, VOCAB, 0, suppress_eos=True), synth_mimi_weights(mc, 0),
                                      args.cpu_utts)), flush=True)
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
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from sopro_amd import hip

    log("building engine")
    tts, cfg, mc, wn, mn = build_engine(device)
    ids, ref_tq = make_inputs(rank)
    ref = tts.prepare_reference(ref_tokens_tq=ref_tq)  # per-voice, outside the timed region (README "precalculate" flow)
    refs = [ref] * BATCH
    kw = dict(max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, text_ids=ids)

    job = dict(texts=[""] * BATCH, refs=refs, **kw)

    def check(out):
        assert all(o.shape[-1] == FRAMES * 1920 for o in out)

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
            for out in pipe.run([job] * n, timings=timings):
                check(out)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("warmup")
    # every lane runs a shape eagerly once (scratch allocation) and records its launch sequences on the second pass
    warm = max(args.warmup, 2 * args.lanes if pipe is not None else 2)
    run_steps(warm)
    fence()
    log("timed steps")
    phases = {}
    t0 = time.perf_counter()
    run_steps(args.steps, phases)
    fence()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f} s for {args.steps} steps; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if world > 1:
        t = torch.tensor([dt], device="cpu" if share_gpu else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- instrumented repeat of the same steps (same lanes, streams and CU partitions): HIP events around every GEMM /
    # attention launch and AR frame replay, recorded on the stream of the launch.  The recorded NAR / Mimi launch sequences
    # are issued eagerly here so that their launches are visible one by one; the timed region above is not instrumented.
    prof = hip.Profiler()
    nprof = max(0, min(args.profile_steps, args.steps))
    if nprof > 0:
        hip.set_profiler(prof)
        run_steps(nprof)
        fence()
        hip.set_profiler(None)

    # ---- roofline of the dominant kernel family (HIP events recorded on the engine streams during the timed steps)
    fam = prof.summary()
    pmc = {}
    try:  # HBM bytes per launch from the rocprofv3 PMC passes of this command (tools/pmc_summary.py -> profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))["families"]
    except Exception:  # noqa: BLE001
        pmc = {}
    try:  # matrix-core busy share from the SQ_VALU_MFMA_BUSY_CYCLES pass (tools/pmc_summary.py --mfma; whole chip per kernel)
        busy = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_mfma_busy.json")))["families"]
    except Exception:  # noqa: BLE001
        busy = {}
    roof, roof_ar = None, None
    share = (256 - args.ar_cus * (1 if args.ar_shared else args.ar_parts)) / 256.0 if args.lanes > 1 else 1.0  # CUs of the bulk partition
    roof_split = None
    if "gemm_bf16x3_kernel" in fam:
        f = fam["gemm_bf16x3_kernel"]
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
        roof_split = {"kernel": "gemm_bf16x3_kernel (Mimi decoder contractions; v_mfma_f32_32x32x16_bf16, 3 passes per product)",
                      "bound": "mfma", "achieved": round(ach, 3), "peak": round(PEAK_BF16_MFMA_TFLOPS * share, 2), "unit": "TFLOP/s",
                      "frac": round(ach / (PEAK_BF16_MFMA_TFLOPS * share), 5), "cu_share": share,
                      "peak_three_pass": round(PEAK_BF16_MFMA_TFLOPS * share / 3.0, 2),
                      "frac_three_pass": round(ach / (PEAK_BF16_MFMA_TFLOPS * share / 3.0), 5),
                      "traffic": pmc.get("gemm_bf16x3_kernel", {}).get("traffic_bytes_per_launch"), "launches": f["launches"],
                      "mfma_busy_pmc": busy.get("gemm_bf16x3_kernel", {}).get("mfma_busy_share_at_2p4ghz"),
                      "avg_launch_us": round(f["ms"] / max(1, f["launches"]) * 1e3, 2),
                      "algorithmic_flops_per_launch": round(f["flops"] / max(1, f["launches"])),
                      "note": "achieved = algorithmic (fp32-equivalent) flops / launch time; each product costs three bf16 MFMA passes"}
    roof_x6 = None
    if "gemm_bf16x6_kernel" in fam:
        f = fam["gemm_bf16x6_kernel"]
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
        roof_x6 = {"kernel": "gemm_bf16x6_kernel (NAR contractions; v_mfma_f32_32x32x16_bf16, 6 passes per product, 24-bit operands)",
                   "bound": "mfma", "achieved": round(ach, 3), "peak": round(PEAK_BF16_MFMA_TFLOPS * share, 2), "unit": "TFLOP/s",
                   "frac": round(ach / (PEAK_BF16_MFMA_TFLOPS * share), 5), "cu_share": share,
                   "peak_six_pass": round(PEAK_BF16_MFMA_TFLOPS * share / 6.0, 2),
                   "frac_six_pass": round(ach / (PEAK_BF16_MFMA_TFLOPS * share / 6.0), 5),
                   "traffic": pmc.get("gemm_bf16x6_kernel", {}).get("traffic_bytes_per_launch"), "launches": f["launches"],
                   "mfma_busy_pmc": busy.get("gemm_bf16x6_kernel", {}).get("mfma_busy_share_at_2p4ghz"),
                   "avg_launch_us": round(f["ms"] / max(1, f["launches"]) * 1e3, 2),
                   "algorithmic_flops_per_launch": round(f["flops"] / max(1, f["launches"])),
                   "note": "achieved = algorithmic (fp32-equivalent) flops / launch time; each product costs six bf16 MFMA passes"}
    if "gemm_f32_kernel" in fam:
        f = fam["gemm_f32_kernel"]
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
        roof = {"kernel": "gemm_f32_kernel (all tile shapes; v_mfma_f32_32x32x2_f32)", "bound": "mfma", "achieved": round(ach, 3),
                "peak": round(PEAK_F32_MFMA_TFLOPS * share, 2), "unit": "TFLOP/s", "frac": round(ach / (PEAK_F32_MFMA_TFLOPS * share), 5),
                "cu_share": share, "peak_full_chip": PEAK_F32_MFMA_TFLOPS, "frac_full_chip": round(ach / PEAK_F32_MFMA_TFLOPS, 5),
                "traffic": pmc.get("gemm_f32_kernel", {}).get("traffic_bytes_per_launch"), "launches": f["launches"],
                "avg_launch_us": round(f["ms"] / max(1, f["launches"]) * 1e3, 2),
                "algorithmic_flops_per_launch": round(f["flops"] / max(1, f["launches"]))}
    if "ar_step_graph" in fam:
        f = fam["ar_step_graph"]
        per_launch_ms = f["ms"] / max(1, f["launches"])
        bytes_step = ar_step_bytes(BATCH, TEXT_LEN)
        ach = bytes_step / (per_launch_ms * 1e-3) / 1e9
        tr = None
        if pmc:
            per_frame = {"skinny_kernel": 19, "xattn_step_kernel": 3, "ar_sample_kernel": 1}
            tr = round(sum(pmc.get(k, {}).get("traffic_bytes_per_launch", 0) * n for k, n in per_frame.items()))
        roof_ar = {"kernel": "AR frame (hipGraph of 23 launches: skinny_kernel x19, xattn_step_kernel x3, ar_sample_kernel)", "bound": "hbm",
                   "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 5),
                   "traffic": tr, "launches": f["launches"], "avg_launch_us": round(per_launch_ms * 1e3, 2),
                   "algorithmic_bytes_per_launch": bytes_step}
    # `roofline` = the compute kernel family with the largest share of a step; the others follow as roofline_more
    cands = [(fam[k]["ms"], r) for k, r in (("gemm_f32_kernel", roof), ("gemm_bf16x3_kernel", roof_split),
                                                 ("gemm_bf16x6_kernel", roof_x6)) if r is not None]
    cands.sort(key=lambda t: -t[0])
    roof = cands[0][1] if cands else None
    roof_more = [r for _, r in cands[1:]] + ([roof_ar] if roof_ar is not None else [])
    for r in [roof] + roof_more:
        if r is not None:
            r["measured"] = f"HIP events on the launch stream over an instrumented repeat of {nprof} steps right after the timed region"
    families = {k: {"ms_per_step": round(v["ms"] / max(1, nprof), 3), "launches_per_step": v["launches"] // max(1, nprof),
                    "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 3) if v["flops"] else None} for k, v in fam.items()}

    if pipe is not None:
        pipe.close()  # the latency leg below runs on the whole chip

    # ---- p50 time-to-first-audio of stream(), batch 1 (BASELINE.json configs[2]); outside the timed steps
    ttfa = None
    if rank == 0 and args.ttfa_runs > 0:
        log("ttfa")
        lat = []
        for i in range(args.ttfa_runs + 3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            it = tts.stream("", ref=ref, max_frames=FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True, chunk_frames=6,
                            text_ids=ids[i % BATCH])
            first = next(it)
            torch.cuda.synchronize()
            if i >= 3:
                lat.append((time.perf_counter() - t1) * 1e3)
            del it, first
        ttfa = float(np.percentile(lat, 50))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess

        log("cpu baseline (oracle, child process, <= 150 s)")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-utts", str(args.cpu_utts), "--frames", str(FRAMES),
                                "--batch", str(BATCH)],
                               capture_output=True, text=True, timeout=150, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001  (a missing baseline must not void the GPU measurement)
            log(f"cpu baseline failed: {e!r}")
            cpu = None

    if rank == 0:
        audio_sec = world * args.steps * BATCH * FRAMES * FRAME_SEC
        line = {
            "metric": "audio_seconds_per_second", "value": round(audio_sec / dt, 2), "unit": "audio-s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "warmup_run": warm, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Sopro-135M synthesize, {BATCH} utterances x {FRAMES} frames per GPU ({'BASELINE configs[1]' if (BATCH, FRAMES) == (32, 200) else 'non-default shape'}), "
                                   f"S={TEXT_LEN} text tokens, {REF_FRAMES}-frame reference voice prepared outside the timed region, "
                                   "top_p=0.9 T=1.05 anti_loop (reference defaults), synthetic weights with EOS suppressed",
                       "batch_per_gpu": BATCH, "frames": FRAMES, "parallelism": f"replicas x{world} (utterance sharding, no collective)",
                       "lanes_per_gpu": args.lanes,
                       "pipelining": (f"{args.lanes} engines per GPU share the weights: up to {args.ar_parts} AR phases at a time on "
                                      f"{'one shared partition' if args.ar_shared else 'partitions'} of {args.ar_cus} CUs while conditioning, NAR and Mimi "
                                      f"decode of other batches run on the other {int(round(256 * share))} CUs (hipExtStreamCreateWithCUMask); NAR and Mimi "
                                      "launch sequences are recorded hipGraphs") if args.lanes > 1 else "none"},
            "dtype_detail": "fp32 tensors and accumulation everywhere; conditioning + AR on v_mfma_f32_*_f32; NAR contractions with operands split into "
                            "three bf16 pieces (24 mantissa bits, 6 MFMA passes); Mimi decoder contractions with two pieces (16 bits, 3 passes)",
            "phase_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phases.items()},
            "kernel_families": families,
            "ttfa_ms_p50": None if ttfa is None else round(ttfa, 3),
            "roofline": roof, "roofline_more": roof_more, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
