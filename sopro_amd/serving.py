"""Batched admission in front of the engine (the serving loop's engine side).

The reference's demo server serialises requests behind one global lock (demo/server.py:56, 224, 241): one utterance on
the device at a time.  Here concurrent callers ``submit`` requests; a scheduler groups compatible ones (same sampling
parameters and frame budget: those are per-launch constants of the AR graph) into batches of up to ``max_batch`` rows,
waiting at most ``max_wait_ms`` for a batch to fill, and the batches flow through the lanes of a
``PipelinedSynthesizer`` (generation of one batch overlaps refinement / decoding of others).  Text lengths, reference
voices and end-of-speech times may differ inside a batch.  Transport (HTTP, framing into responses) stays outside:
``sopro_amd.wire`` has the byte formats.
"""
from __future__ import annotations

import atexit
import queue
import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch

from .model import PreparedReference


@dataclass
class _Request:
    text_ids: torch.Tensor
    ref: PreparedReference
    key: Tuple
    future: Future
    t_submit: float


class SynthesisService:
    def __init__(self, tts, *, max_batch: int = 32, max_wait_ms: float = 4.0, lanes: int = 4, ar_cus: int = 64, ar_parts: int = 2,
                 ar_shared: bool = True, mode: str = "batch", **continuous_kw):
        """``mode="batch"``: requests with equal parameters are grouped into batches for the lanes of a PipelinedSynthesizer.
        ``mode="continuous"``: frame-level admission (``ContinuousSynthesizer``; extra keywords go to it): parameters, frame
        budgets and end-of-speech times may all differ between neighbouring slots."""
        from .pipeline import PipelinedSynthesizer

        self.tts = tts
        self._closed = False
        self.stats = {"requests": 0, "batches": 0, "rows": 0}
        self.engine = None
        if mode == "continuous":
            from .continuous import ContinuousSynthesizer

            kw = dict(slots=max_batch, ar_cus=ar_cus, generators=max(1, ar_parts))
            kw.update(continuous_kw)
            self.engine = ContinuousSynthesizer(tts, **kw)
            self.engine.start()
            self.pipe, self._threads = None, []
            return
        if mode != "batch":
            raise ValueError("mode must be 'batch' or 'continuous'")
        self.max_batch, self.max_wait = int(max_batch), float(max_wait_ms) * 1e-3
        self.pipe = PipelinedSynthesizer(tts, lanes=lanes, ar_cus=ar_cus, ar_parts=ar_parts, ar_shared=ar_shared) if lanes > 1 else None
        self._lanes = self.pipe.lanes if self.pipe is not None else [tts]
        self._inbox: "queue.Queue[Optional[_Request]]" = queue.Queue()
        self._batches: "queue.Queue[Optional[List[_Request]]]" = queue.Queue(maxsize=2 * len(self._lanes))
        self._threads = [threading.Thread(target=self._schedule, name="sopro-sched", daemon=True)]
        for i, lane in enumerate(self._lanes):
            self._threads.append(threading.Thread(target=self._work, args=(lane, i), name=f"sopro-lane{i}", daemon=True))
        for t in self._threads:
            t.start()
        atexit.register(self.close)  # worker threads must not be inside the HIP runtime when the interpreter tears it down

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ client side
    def submit(self, text: str, ref: PreparedReference, *, max_frames: int = 400, top_p: float = 0.9, temperature: float = 1.05,
               anti_loop: bool = True, style_strength: Optional[float] = None, min_gen_frames: Optional[int] = None,
               text_ids: Optional[torch.Tensor] = None) -> "Future[torch.Tensor]":
        """Queue one utterance; the future resolves to the waveform ``[1, 1, N]`` on the device (``synthesize``'s result)."""
        if self._closed:
            raise RuntimeError("service is closed")
        if self.engine is not None:
            self.stats["requests"] += 1
            return self.engine.submit(text=text, text_ids=text_ids, ref=ref, max_frames=max_frames, top_p=top_p, temperature=temperature,
                                      anti_loop=anti_loop, style_strength=style_strength, min_gen_frames=min_gen_frames)
        ids = text_ids if text_ids is not None else self.tts.encode_text(text)
        if int(ids.numel()) == 0:
            raise ValueError("empty text")
        ss = float(style_strength if style_strength is not None else self.tts.cfg.style_strength)
        key = (int(max_frames), float(top_p), float(temperature), bool(anti_loop), ss, min_gen_frames)
        fut: Future = Future()
        self._inbox.put(_Request(ids, ref, key, fut, time.perf_counter()))
        return fut

    def synthesize(self, text: str, ref: PreparedReference, **kw) -> torch.Tensor:
        return self.submit(text, ref, **kw).result()

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        atexit.unregister(self.close)
        if self.engine is not None:
            self.engine.close()
            return
        self._inbox.put(None)
        for t in self._threads:
            t.join()
        if self.pipe is not None:
            self.pipe.close()

    # ------------------------------------------------------------------ scheduler: group compatible requests
    def _schedule(self) -> None:
        pending: Dict[Tuple, List[_Request]] = {}
        stop = False
        while not stop or pending:
            deadline = min((rs[0].t_submit + self.max_wait for rs in pending.values()), default=None)
            timeout = None if deadline is None else max(0.0, deadline - time.perf_counter())
            try:
                r = self._inbox.get(timeout=timeout) if not stop else None
                if r is None and not stop:
                    stop = True
                elif r is not None:
                    pending.setdefault(r.key, []).append(r)
                    while True:  # drain whatever else is already queued
                        try:
                            r2 = self._inbox.get_nowait()
                        except queue.Empty:
                            break
                        if r2 is None:
                            stop = True
                        else:
                            pending.setdefault(r2.key, []).append(r2)
            except queue.Empty:
                pass
            now = time.perf_counter()
            for key in list(pending):
                rs = pending[key]
                while len(rs) >= self.max_batch:
                    self._batches.put(rs[: self.max_batch])
                    del rs[: self.max_batch]
                if rs and (stop or now >= rs[0].t_submit + self.max_wait):
                    self._batches.put(list(rs))
                    rs.clear()
                if not rs:
                    del pending[key]
        for _ in self._lanes:
            self._batches.put(None)

    # ------------------------------------------------------------------ lanes: run batches
    def _work(self, lane, idx: int) -> None:
        locks = (self.pipe.ar_locks[idx % self.pipe.ar_parts], self.pipe.bulk_lock) if self.pipe is not None else None
        with torch.cuda.stream(lane.model.stream):
            while True:
                batch = self._batches.get()
                if batch is None:
                    return
                mf, top_p, temp, anti, ss, mg = batch[0].key
                try:
                    out = lane.synthesize_batch([""] * len(batch), [r.ref for r in batch], max_frames=mf, top_p=top_p, temperature=temp,
                                                anti_loop=anti, style_strength=ss, min_gen_frames=mg, text_ids=[r.text_ids for r in batch],
                                                phase_locks=locks)
                    self.stats["requests"] += len(batch)
                    self.stats["batches"] += 1
                    self.stats["rows"] += len(batch)
                    for r, w in zip(batch, out):
                        r.future.set_result(w)
                except BaseException as e:  # noqa: BLE001
                    for r in batch:
                        if not r.future.done():
                            r.future.set_exception(e)
