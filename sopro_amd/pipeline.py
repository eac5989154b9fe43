"""Two-lane pipelining of synthesis batches on one MI355X (BASELINE.json configs[4]: AR loop and codec decoder
overlapped on two HIP streams).

A batch has a latency-bound phase (conditioning + ~200 replays of the 23-launch AR frame graph: the GPU is mostly
idle, every launch waits for the previous one) and a throughput-bound phase (NAR refinement + Mimi decode: large
fp32 MFMA contractions that fill the chip).  Run back to back the first phase wastes the chip; run concurrently on
ordinary streams the AR launches queue behind thousands of resident GEMM workgroups (5x slower per frame,
measured).  So the chip is partitioned with CU masks (`hipExtStreamCreateWithCUMask`): the AR stream owns
`ar_cus` CUs, the bulk stream the rest, and two engines ("lanes") sharing the device weights alternate:
lane A generates batch k+1 while lane B refines / decodes batch k.  There is no data-path communication between
lanes; results are identical to the sequential path (same kernels, same order per batch).
"""
from __future__ import annotations

import os
import threading
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from . import hip


class PipelinedSynthesizer:
    def __init__(self, tts, lanes: int = 2, ar_cus: int = 64, ar_parts: int = 1, ar_shared: bool = False, bulk_slots: int = 1,
                 prep_on_ar: Optional[bool] = None):
        """``ar_parts`` AR partitions of ``ar_cus`` CUs each (the AR phase is launch-latency bound, so independent
        partitions generate independent batches concurrently); the remaining CUs form the one bulk partition.
        Lane i generates on partition i % ar_parts.  With ``ar_shared`` the partitions are ONE CU range of ``ar_cus`` CUs
        that ``ar_parts`` AR phases use at the same time (their short kernels interleave on the same CUs)."""
        self.device = tts.device
        self.unpartitioned = int(ar_cus) <= 0
        if self.unpartitioned:
            self._init_unpartitioned(tts, int(lanes), max(1, int(ar_parts)), bulk_slots)
            return
        if getattr(tts.model, "_driver", None) is not None:
            raise RuntimeError("this engine is already driven by a " + type(tts.model._driver).__name__ + " (close it first): "
                               "the schedulers re-point the engine's streams at CU partitions")
        total = hip.device_info(self.device.index or 0)["cus"]
        ar_parts = max(1, int(ar_parts))
        n_ar = ar_cus if ar_shared else ar_cus * ar_parts
        if not (0 < n_ar < total):
            raise ValueError("the AR partitions must leave CUs for the bulk phase")
        self.lanes = []
        self._saved = (tts.model.stream, tts.model.bulk_stream, tts.codec.stream, tts.model.prep_stream)
        self._streams = []
        bulk0 = n_ar
        # CU ranges, not arbitrary masks: mask bit i is slot i // 8 of XCD i % 8, workgroups are dealt to ALL eight XCDs whatever the
        # mask says, and an XCD whose slots are all masked out runs the stream on all of its CUs (profiles/r03_mask_census.txt) -
        # a partition can take fewer CUs of every XCD, never fewer XCDs.
        mk = lambda lo, n: hip.cu_range_stream(lo, n, self.device)  # noqa: E731
        for i in range(int(lanes)):
            lane = tts if i == 0 else tts.clone_lane()
            # (one stream set per lane, not per partition: sharing the AR stream of a lock, one refinement / decode stream and
            # one conditioning stream between the lanes - 6 hardware queues instead of 12 - measured 31.3 against 29.5 ms per
            # step: the conditioning of one lane then queues behind another lane's decode)
            lane.model.stream = mk(0 if ar_shared else (i % ar_parts) * ar_cus, ar_cus)
            lane.model.bulk_stream = mk(bulk0, total - bulk0)
            lane.model.prep_stream = lane.model.bulk_stream  # idle while this lane generates; GEMM-shaped preparation belongs there
            if prep_on_ar if prep_on_ar is not None else hip.dev_env("SOPRO_PREP_ON_AR", "0") == "1":
                # (with four jobs per pass the generation partition has slack and the throughput partition is the bound: the
                # per-pass preparation - conditioning, text folding - on a stream of its own over the GENERATION partition's CUs)
                lane.model.prep_stream = mk(0 if ar_shared else (i % ar_parts) * ar_cus, ar_cus)
                self._streams.append(lane.model.prep_stream)
            self._streams += [lane.model.stream, lane.model.bulk_stream]
            lane.codec.stream = lane.model.bulk_stream
            lane.model._ar_cache.clear()  # recorded graphs belong to the stream they were captured on
            lane.model._nar_graphs.clear()
            lane.codec._graphs.clear()
            if hip.dev_env("SOPRO_AR_TILES_WIDE", "1x2") not in ("", "0"):
                lane.model.ar_tiles_wide = hip.dev_env("SOPRO_AR_TILES_WIDE", "1x2")  # coalesced (64-row) frames on the small partition
            self.lanes.append(lane)
        # An empty pipeline has nothing on the throughput partition yet: the first AR phase of each partition lock gets an equal
        # share of the WHOLE chip (the recorded frame graph replays on any stream), which shortens the fill of the pipeline.
        # These are CU-masked streams too (disjoint ranges): an ordinary stream generating beside a CU-masked one was
        # measured at 390-830 us per frame for both (tools/ar_concurrency_probe.py, "64-CU partition + whole chip"), while
        # two disjoint halves give the same 136 us per frame as two ordinary streams.
        share = max(32, (total // ar_parts) // 32 * 32)  # whole multiples of 32 CUs (see the partition-size note in DESIGN.md)
        self._full = [mk((i % ar_parts) * share, share) for i in range(int(lanes))]
        self._streams += self._full
        # ... and a pipeline that is running dry has nothing left for the generation partition: once every job of a run is past
        # its AR phase, the remaining refinement / decode phases take the whole chip (CU-masked streams over all CUs: an
        # ordinary stream beside masked ones is the slow combination noted above).  SOPRO_DRAIN_WHOLE=0: stay on the partition.
        self._whole = None
        if hip.dev_env("SOPRO_DRAIN_WHOLE", "1") != "0":
            # (one stream for all lanes while the throughput slot is exclusive: its phases are serial anyway)
            own = [mk(0, total) for _ in range(int(lanes) if bulk_slots > 1 else 1)]
            self._whole = [own[i % len(own)] for i in range(int(lanes))]
            self._streams += own
        self.ar_locks = [threading.Lock() for _ in range(ar_parts)]
        self.bulk_lock = threading.Lock() if bulk_slots <= 1 else threading.BoundedSemaphore(int(bulk_slots))
        self._share_decoder_scratch(bulk_slots)
        self.ar_cus, self.ar_parts, self.bulk_cus = ar_cus, ar_parts, total - bulk0
        tts.model._driver = self
        # (The lane threads spend most of their time WAITING - AR stop polls one chunk behind the launches, the end of a refinement /
        # decode phase.  With the runtime's default spin wait that is one busy core per lane: r03 measured 3.6 cores per rank.  The
        # remedy is hip.set_host_wait(True) - and it is a PROCESS-level decision that has to be taken before the first stream of
        # the device exists: see its docstring; bench.py and a serving process take it at start-up.)

    def _init_unpartitioned(self, tts, lanes: int, ar_parts: int, bulk_slots: int) -> None:
        """``ar_cus <= 0``: no CU masks.  Every stream may use the whole chip; the long contraction kernels of the throughput
        phases are capped at one workgroup per CU (``hip.set_lds_floor``), so every CU always has wave slots, registers and
        LDS left for the short kernels of the AR frames, which are issued on high-priority streams."""
        if getattr(tts.model, "_driver", None) is not None:
            raise RuntimeError("this engine is already driven by a " + type(tts.model._driver).__name__ + " (close it first)")
        self.lanes = []
        self._saved = (tts.model.stream, tts.model.bulk_stream, tts.codec.stream, tts.model.prep_stream)
        self._streams = []
        hip.set_lds_floor(int(hip.dev_env("SOPRO_LDS_FLOOR_KB", "84")) * 1024)
        for i in range(lanes):
            lane = tts if i == 0 else tts.clone_lane()
            lane.model.stream = torch.cuda.Stream(device=self.device, priority=-1)
            lane.model.bulk_stream = torch.cuda.Stream(device=self.device, priority=0)
            lane.model.prep_stream = lane.model.bulk_stream
            lane.codec.stream = lane.model.bulk_stream
            lane.model._ar_cache.clear()
            lane.model._nar_graphs.clear()
            lane.codec._graphs.clear()
            self.lanes.append(lane)
        self._full = [lane.model.stream for lane in self.lanes]
        self._whole = None
        self.ar_locks = [threading.Lock() for _ in range(ar_parts)]
        self.bulk_lock = threading.Lock() if bulk_slots <= 1 else threading.BoundedSemaphore(int(bulk_slots))
        self._share_decoder_scratch(bulk_slots)
        self.ar_cus, self.ar_parts, self.bulk_cus = 0, ar_parts, hip.device_info(self.device.index or 0)["cus"]
        tts.model._driver = self

    def _share_decoder_scratch(self, bulk_slots: int) -> None:
        """One throughput slot = one refinement / decode phase at a time (``bulk_lock``; ``prepare`` runs the lanes one by one): the
        lanes decode in lane 0's scratch buffers instead of holding one set each (SOPRO_SHARE_SCRATCH=0: one set per lane)."""
        if int(bulk_slots) <= 1 and hip.dev_env("SOPRO_SHARE_SCRATCH", "1") != "0":
            for lane in self.lanes[1:]:
                lane.codec.share_scratch(self.lanes[0].codec.ws)

    def close(self) -> None:
        """Drop the extra lanes, destroy the CU-masked streams (and the graphs recorded on them) and give lane 0
        (the caller's engine) its full-chip streams back."""
        if not self.lanes:
            return
        if self.unpartitioned:
            hip.set_lds_floor(0)
        torch.cuda.synchronize(self.device)
        for lane in self.lanes:
            lane.model._ar_cache.clear()  # hipGraphExecDestroy now, not at interpreter shutdown
            lane.model._nar_graphs.clear()
            lane.codec._graphs.clear()
            lane.model.ws.clear()
            lane.codec.ws.clear()
        lane0 = self.lanes[0]
        lane0.codec.ws.on_clear[:] = [lane0.codec._graphs.clear]  # (the dropped lanes' recorded calls are gone)
        lane0.model._driver = None
        lane0.model.ar_tiles_wide = None
        lane0.model.stream, lane0.model.bulk_stream, lane0.codec.stream, lane0.model.prep_stream = self._saved
        self.lanes = []
        torch.cuda.synchronize(self.device)
        import gc

        gc.collect()  # Graph.__del__ of the dropped plans parks the handles ...
        hip.reap_parked_graphs()  # ... and the device is idle, no recording is open: destroy them before their streams go
        # torch's caching allocator keeps freed blocks per STREAM: the scratch just dropped was allocated on streams that are about to
        # be destroyed, could never be reused by any other stream, and was seen to take the process down later (a SIGSEGV inside
        # torch.empty in the fifth pipeline of one process: profiles/r04_experiments.md).  Hand it back while its streams exist.
        torch.cuda.empty_cache()
        for st in self._streams:
            hip.destroy_stream(st)
        self._streams = []

    _PER_UTT = ("texts", "refs", "text_ids")  # job keys that are per-utterance lists

    def pass_sizes(self, n_jobs: int, coalesce) -> List[int]:
        """How many consecutive jobs each pass of a run takes.  An integer ``coalesce``: that many per pass (the last one what is
        left); a list: explicit sizes (developer sweeps).  ``"auto"``: FOUR jobs per pass once the queue holds four jobs per lane,
        two below that.  The AR frame chain costs little more for 128 rows than for 64 (two phases at a time: 27-30 ms of AR phases
        per 32-utterance job instead of 35), so the generation partition gets slack and the run is bound by refinement + decoding
        alone - measured (profiles/r05_experiments.md section 3): steady state 25.2 k audio-s/s at two per pass, 26.0-26.2 k at
        four / six / eight; the 20-job driver form 23.6-24.0 k at two, 24.6-25.0 k at four (five equal passes on four lanes).
        Mixed sizes lose: a smaller first pass (the throughput partition gets its first work sooner) was measured with every lane
        warmed up on every shape - [1, 2 x 9, 1] 23.6-23.7 k, [2, 4, 4, 4, 4, 2] 24.3-24.4 k, [3, 4, 4, 4, 5] 24.1-24.3 k,
        [2, 2, 4, 4, 4, 4] 23.4 k against 24.9-25.0 k for [4] x 5 on the same box."""
        if isinstance(coalesce, (list, tuple)):  # explicit sizes (developer sweeps); what they leave over runs n-at-a-time with their last size
            sizes, left = [], n_jobs
            for v in coalesce:
                if left <= 0:
                    break
                sizes.append(min(max(1, int(v)), left))
                left -= sizes[-1]
            while left > 0:
                sizes.append(min(sizes[-1], left))
                left -= sizes[-1]
            return sizes
        if coalesce != "auto":
            n = max(1, int(coalesce))
        else:
            n = 4 if n_jobs >= 4 * len(self.lanes) else 2
        return [min(n, n_jobs - i) for i in range(0, n_jobs, n)]

    def prepare(self, job: Dict[str, Any], sizes: Sequence[int] = (1, 2)) -> None:
        """Deterministic warm-up: EVERY lane runs a pass of each size in ``sizes`` (that many copies of ``job`` coalesced) twice - the
        first time eagerly (scratch allocation), the second time recording its launch sequences - one lane at a time, on its own
        partition streams.  A scheduler that mixes pass sizes (``coalesce="auto"``) cannot rely on the run itself for that: which
        lane meets which shape first is a race (round 4's ramp-up experiment fell to 14.7 k in one run of four that way)."""
        for lane in self.lanes:
            with torch.cuda.stream(lane.model.stream):
                for sz in sorted({int(v) for v in sizes}):
                    merged = self._coalesce([job] * sz, sz)[0][1] if sz > 1 else job
                    for _ in range(2):
                        lane.synthesize_batch(**merged)
        torch.cuda.synchronize(self.device)

    def _coalesce(self, jobs: Sequence[Dict[str, Any]], n, ramp: bool = False, sizes: Optional[Sequence[int]] = None):
        """Groups of up to ``n`` CONSECUTIVE jobs with equal sampling parameters become one pass over their concatenated
        utterances (the AR frame chain costs nearly the same for 64 rows as for 32: profiles/r03_experiments.md).  Every
        utterance keeps the sampler stream it has in its own job - that job's nonce and its index within the job (``nonces`` /
        ``row_ids`` of synthesize_batch) - so results are bit-identical to running the jobs one by one FOR SEEDED JOBS; a job
        without a ``seed`` takes the next nonce of the process-wide run counter here, in job order, which is not the order the
        passes later execute in (a seedless job is "a new take" either way)."""
        groups: List[List[int]] = []
        caps = list(sizes) if sizes is not None else None  # explicit pass sizes (pass_sizes): group g takes at most caps[g] jobs
        n = max(caps) if caps else int(n)
        for i, j in enumerate(jobs):
            head = jobs[groups[-1][0]] if groups else None
            # ``ramp`` (SOPRO_PIPE_RAMP=1; off by default): the first pass of a run stays a single job - its conditioning and
            # generation are the shortest possible, so the throughput partition gets its first phase ~20 ms earlier.  Measured
            # (profiles/r04_experiments.md): within the run-to-run spread, and the extra pass shape (32 rows next to 64) can land on
            # a lane that has not recorded its launch sequences yet - one run in four fell to 14.7 k
            cap = 1 if (ramp and len(groups) == 1) else n
            if caps is not None and groups:
                cap = caps[len(groups) - 1] if len(groups) - 1 < len(caps) else n
            same = bool(groups) and len(groups[-1]) < cap and all(
                head.get(k) == j.get(k) for k in (set(j) | set(head)) - set(self._PER_UTT) - {"seed"})
            # ... and the same per-utterance lists present (one job with `texts`, its neighbour with `text_ids` do not merge)
            same = same and all((head.get(k) is None) == (j.get(k) is None) for k in self._PER_UTT)
            if same:
                groups[-1].append(i)
            else:
                groups.append([i])
        passes = []
        model = self.lanes[0].model
        for g in groups:
            if len(g) == 1:
                passes.append((g, jobs[g[0]], None))
                continue
            merged = {k: v for k, v in jobs[g[0]].items() if k not in self._PER_UTT and k != "seed"}
            sizes = [len(jobs[i]["refs"]) for i in g]
            for k in self._PER_UTT:
                if jobs[g[0]].get(k) is not None:
                    merged[k] = [u for i in g for u in jobs[i][k]]
            merged["nonces"] = [nn for i, sz in zip(g, sizes) for nn in [model.next_nonce(jobs[i].get("seed"))] * sz]
            merged["row_ids"] = [r for sz in sizes for r in range(sz)]
            passes.append((g, merged, sizes))
        return passes

    def run(self, jobs: Sequence[Dict[str, Any]], timings: Optional[Dict[str, float]] = None, coalesce=1) -> List[Any]:
        """Each job is the keyword dict of ``SoproTTS.synthesize_batch``; results come back in job order.  ``coalesce`` > 1:
        consecutive compatible jobs are generated, refined and decoded together (see ``_coalesce``); ``"auto"``: pass sizes chosen
        from the queue depth (``pass_sizes``; call ``prepare`` first so that every lane has recorded every pass shape)."""
        if (coalesce == "auto" or isinstance(coalesce, (list, tuple)) or int(coalesce) > 1) and len(jobs) > 1:
            if coalesce == "auto" or isinstance(coalesce, (list, tuple)):
                passes = self._coalesce(jobs, 0, sizes=self.pass_sizes(len(jobs), coalesce))
            else:
                passes = self._coalesce(jobs, int(coalesce), ramp=len(jobs) > 2 * int(coalesce) and hip.dev_env("SOPRO_PIPE_RAMP", "0") == "1")
            outs = self.run([p[1] for p in passes], timings=timings)
            results: List[Any] = [None] * len(jobs)
            for (g, _m, sizes), out in zip(passes, outs):
                if sizes is None:
                    results[g[0]] = out
                else:
                    o = 0
                    for i, sz in zip(g, sizes):
                        results[i] = out[o:o + sz]
                        o += sz
            return results
        results = [None] * len(jobs)
        errors: List[BaseException] = []
        nxt = [0]
        pick = threading.Lock()

        # An empty pipeline has nothing on the throughput partition yet, so the FIRST AR phase of each partition lock runs on its
        # share of the whole chip (`self._full`: 136 instead of 255-270 us per frame), which shortens the fill of the pipeline.
        # Which phase that is gets decided when the lock is TAKEN, not by job index: a fill phase that lost the race for its
        # lock and then ran beside a partition-bound one was measured at 400-750 us per frame for both
        # (profiles/r02_experiments.md, "start-up race") - the slow mode of one bench run in three.
        ar_started = [0] * self.ar_parts
        ar_finished = [0]

        class _ArSlot:
            def __init__(slot, lane, part, lane_idx):
                slot.lane, slot.part, slot.lane_idx = lane, part, lane_idx

            def __enter__(slot):
                self.ar_locks[slot.part].acquire()
                slot.masked = slot.lane.model.stream
                with pick:
                    first = ar_started[slot.part] == 0 and ar_finished[0] == 0
                    ar_started[slot.part] += 1
                slot.fill = first and not self.unpartitioned
                if slot.fill:
                    slot.lane.model.stream = self._full[slot.lane_idx]

            def __exit__(slot, *exc):
                slot.lane.model.stream = slot.masked
                with pick:
                    ar_finished[0] += 1
                self.ar_locks[slot.part].release()
                return False

        # Conditioning phases take turns IN JOB ORDER.  When a run starts, every lane conditions its first pass at the same moment
        # on the same CUs: each took 11-12 ms instead of 3 (lane trace, r05 call 4), so the first generation phases - whose end is
        # when the throughput partition gets its first work - started 8-9 ms late.  In steady state the phases rarely meet.
        cond_turn = [0]
        cond_cv = threading.Condition()

        class _CondGate:
            def __init__(gate, idx):
                gate.idx = idx

            def __enter__(gate):
                with cond_cv:
                    while cond_turn[0] < gate.idx and not errors:
                        cond_cv.wait(timeout=0.05)

            def __exit__(gate, *exc):
                with cond_cv:
                    cond_turn[0] = max(cond_turn[0], gate.idx + 1)
                    cond_cv.notify_all()
                return False

        class _BulkSlot:
            def __init__(slot, lane, lane_idx):
                slot.lane, slot.lane_idx = lane, lane_idx

            def __enter__(slot):
                self.bulk_lock.acquire()
                slot.saved = None
                with pick:
                    dry = ar_finished[0] >= len(jobs)  # every job of this run is past its AR phase
                if dry and self._whole is not None:
                    slot.saved = (slot.lane.model.bulk_stream, slot.lane.codec.stream)
                    slot.lane.model.bulk_stream = slot.lane.codec.stream = self._whole[slot.lane_idx]
                    self.drain_jobs += 1

            def __exit__(slot, *exc):
                if slot.saved is not None:
                    slot.lane.model.bulk_stream, slot.lane.codec.stream = slot.saved
                self.bulk_lock.release()
                return False

        def worker(lane, ar_lock, lane_idx):
            # the worker's current stream is the lane's own (never the NULL stream, which would serialise the lanes)
            bulk_slot = _BulkSlot(lane, lane_idx)
            with torch.cuda.stream(lane.model.stream):
                while True:
                    with pick:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(jobs) or errors:
                        return
                    tj = {} if timings is not None else None
                    t_job = time.perf_counter()
                    try:
                        results[i] = lane.synthesize_batch(phase_locks=(ar_lock, bulk_slot, _CondGate(i)), timings=tj, **jobs[i])
                    except BaseException as e:  # noqa: BLE001
                        errors.append(e)
                        with cond_cv:
                            cond_cv.notify_all()
                        return
                    if tj is not None:
                        with pick:
                            for k, v in tj.items():
                                if not k.startswith("_"):  # "_..." = absolute time stamps for the trace, not phase sums
                                    timings[k] = timings.get(k, 0.0) + v
                            self.trace.append((i, lane_idx, t_job - t_run, time.perf_counter() - t_run, tj))
                    with pick:
                        self.fill_jobs += [i] if getattr(ar_lock, "fill", False) else []

        n_run = max(1, min(len(self.lanes), len(jobs)))
        import sys
        import time

        self.trace = []  # (job, lane, start s, end s, per-phase seconds) of the last timed run: who was slow, and when
        self.fill_jobs = []  # jobs whose AR phase ran on a pipeline-fill stream (at most one per partition lock and run)
        self.drain_jobs = 0  # refinement / decode phases of this run that had the whole chip (the pipeline was running dry)
        t_run = time.perf_counter()

        swi = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)  # lanes hand the interpreter over between launches; 5 ms hand-over stalls a whole AR poll
        threads = [threading.Thread(target=worker, args=(lane, _ArSlot(lane, i % self.ar_parts, i), i)) for i, lane in enumerate(self.lanes[:n_run])]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        sys.setswitchinterval(swi)
        if errors:
            raise errors[0]
        return results
