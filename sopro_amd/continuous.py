"""Frame-level admission into the batched AR graph ("slot-based admission").

``synthesize_batch`` generates a batch until its LAST utterance has finished: with ragged lengths the rows that reached
end-of-speech (or their frame budget) idle for the rest of the batch.  Here the AR frame graph runs over a fixed set of
slots; every slot has its own time base on the device (``sopro_ar_state.start / row_max / row_params``).  The driver loop

  1. enqueues a chunk of frames for all slots (recorded graph replays),
  2. while the GPU runs them, prepares queued utterances in batches (conditioning, folded text operands) into a ready list,
  3. reads the slot states once per chunk, harvests finished utterances (tokens + their conditioning rows) and installs
     ready utterances in the freed slots (device copies, zeroed ring columns, ``sopro_ar_admit``),
  4. hands finished utterances, in batches of their own, to a worker that refines (NAR) and decodes (Mimi) them on a
     second engine over the same weights.

With ``ar_cus`` the chip is partitioned like ``PipelinedSynthesizer`` does (generation on ``ar_cus`` CUs, everything
GEMM-shaped on the rest).  Per-utterance results are the ones ``synthesize`` gives (greedy decode: identical tokens): a
slot's arithmetic never depends on its neighbours.
"""
from __future__ import annotations

import atexit
import queue
import threading
from collections import deque
from concurrent.futures import Future
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import hip
from .model import RMS_EPS, _ARPlan


class ContinuousSynthesizer:
    def __init__(self, tts, *, slots: int = 32, max_frames: int = 400, max_text: int = 128, poll_every: int = 16,
                 bulk_batch: int = 32, prep_batch: int = 8, ar_cus: Optional[int] = None, generators: int = 1):
        self.tts, self.model = tts, tts.model
        self.slots, self.poll_every = int(slots), int(poll_every)
        self.bulk_batch, self.prep_batch = int(bulk_batch), int(prep_batch)
        self.max_frames = int(max_frames)
        m = self.model
        if getattr(m, "_driver", None) is not None:
            raise RuntimeError("this engine is already driven by a " + type(m._driver).__name__ + " (close it first): "
                               "the schedulers re-point the engine's streams at CU partitions")
        m._driver = self
        self.S_cap = ((int(max_text) + 63) // 64) * 64
        self._own_streams: List[torch.cuda.Stream] = []
        self._saved = (m.stream, m.bulk_stream, m.prep_stream, tts.codec.stream)
        self.bulk_tts = tts.clone_lane()  # refinement + decoding engine: own scratch, own recorded graphs
        if ar_cus is not None:
            total = hip.device_info(m.device.index or 0)["cus"]
            if not (0 < ar_cus < total):
                raise ValueError("ar_cus must leave CUs for the throughput partition")
            m.stream = hip.cu_range_stream(0, ar_cus, m.device)
            m.prep_stream = hip.cu_range_stream(ar_cus, total - ar_cus, m.device)
            bs = hip.cu_range_stream(ar_cus, total - ar_cus, m.device)
            self.bulk_tts.model.stream = self.bulk_tts.model.bulk_stream = self.bulk_tts.model.prep_stream = bs
            self.bulk_tts.codec.stream = bs
            self._own_streams = [m.stream, m.prep_stream, bs]
            m._ar_cache.clear()
        # `generators` slot sets step at the same time on the generation partition (their short kernels interleave, as the
        # concurrent AR phases of PipelinedSynthesizer do); each has its own plan, recorded frame graph and stream
        self.gens: List[Dict[str, Any]] = []
        for gi in range(max(1, int(generators))):
            if gi == 0:
                st = m.stream
            elif ar_cus is not None:
                st = hip.cu_range_stream(0, ar_cus, m.device)
                self._own_streams.append(st)
            else:
                st = torch.cuda.Stream(device=m.device)
            with torch.cuda.stream(st):
                plan = _ARPlan(m, self.slots, self.S_cap, self.max_frames + 1, slots=True)
                hip.ar_init(plan.state)
            plan.ensure_graph(st)
            self.gens.append({"plan": plan, "stream": st})
        self.plan = self.gens[0]["plan"]
        self.stats = {"frames": 0, "slot_frames_used": 0, "utterances": 0, "bulk_batches": 0, "prep_batches": 0}
        self._sh: Optional[Dict[str, Any]] = None
        self._threads: List[threading.Thread] = []
        self._closed = False

    def close(self) -> None:
        m = self.model
        if self._closed:
            return
        self._closed = True
        atexit.unregister(self.close)
        if self._sh is not None:
            self._sh["stop"] = True
            for t in self._threads[1:]:
                t.join()
            self._sh["jobs"].put(None)
            self._threads[0].join()
            self._sh, self._threads = None, []
        torch.cuda.synchronize(m.device)
        for g in self.gens:
            g["plan"].graph = None
            g.pop("snap", None)  # pinned host buffers and events must not outlive the HIP runtime
            g.pop("snap_ev", None)
        self.bulk_tts.model._nar_graphs.clear()
        self.bulk_tts.codec._graphs.clear()
        m.stream, m.bulk_stream, m.prep_stream, self.tts.codec.stream = self._saved
        m._driver = None
        for s in self._own_streams:
            hip.destroy_stream(s)
        self._own_streams = []

    # ------------------------------------------------------------------ preparation of queued utterances (batched)
    def _prepare(self, items: List[Dict[str, Any]]) -> None:
        """Conditioning rows and folded cross-attention operands for a group of requests, on the preparation stream."""
        m, cfg, w, D = self.model, self.model.cfg, self.model.w, self.model.D
        ids = [it["ids"] for it in items]
        mf = max(it["mf"] for it in items)  # rows past an utterance's own budget are never read
        ss = items[0]["ss"]
        prep = m.prepare_conditioning_batch(ids, [it["ref"] for it in items], max_frames=mf, style_strength=ss)
        n, S = len(items), int(prep["txt_seq"].shape[1])
        H = 4
        with m.on_stream(prep=True):
            ts = prep["txt_seq"].contiguous().view(n * S, D)
            nkv = torch.empty(n * S, D, device=m.device)
            kvd = torch.empty(n * S, 2 * D, device=m.device)
            kps, vps = {}, {}
            for i in cfg.ar_xattn_layers:
                pa = f"ar.x_attns.{i}"
                kp = torch.zeros(n, H, self.S_cap, D, device=m.device)
                vp = torch.zeros(n, H, self.S_cap, D, device=m.device)
                hip.ar_fold_text(ts, w[pa + ".nkv.weight"], w[pa + ".kv.w"], w[pa + ".q.wT"], w[pa + ".o.w"], nkv, kvd, kp, vp,
                                 B=n, S=S, S_cap=self.S_cap, D=D, H=H, eps=RMS_EPS)
                kps[i], vps[i] = kp, vp
            ev = torch.cuda.Event()
            ev.record(m.prep_stream)
        for j, it in enumerate(items):
            it.update(cond=prep["cond_ar"][j, : it["mf"] + 1], S=int(prep["text_lens_host"][j]), kp={i: kps[i][j] for i in kps},
                      vp={i: vps[i][j] for i in vps}, ready=ev)
        self.stats["prep_batches"] += 1

    def _admit(self, gen: Dict[str, Any], row: int, it: Dict[str, Any]) -> None:
        """Install a prepared utterance in slot ``row``: device copies + ring columns + sopro_ar_admit, on the AR stream."""
        plan, st = gen["plan"], gen["stream"]
        with torch.cuda.stream(st):
            st.wait_event(it["ready"])
            for t in [it["cond"], *it["kp"].values(), *it["vp"].values()]:
                t.record_stream(st)  # made on the preparation stream, read here
            Tr = it["mf"] + 1
            plan.cond[row, :Tr].copy_(it["cond"])
            plan.klens[row:row + 1].fill_(it["S"])
            for i in plan.kp:
                plan.kp[i][row].copy_(it["kp"][i])
                plan.vp[i][row].copy_(it["vp"][i])
            for r in plan.rings:
                r[:, row].zero_()
            plan.row_max[row:row + 1].fill_(Tr)
            plan.row_params[row].copy_(it["prm"], non_blocking=True)
            nonce = self.model.next_nonce(it.get("seed"))  # a new take per admission unless the request pins its seed
            plan.nonce[row:row + 1].fill_(nonce - (1 << 32) if nonce >= (1 << 31) else nonce)
            hip.ar_admit(plan.state, row)

    # ------------------------------------------------------------------ serving interface
    def start(self) -> None:
        """Spawn the generator threads and the refinement / decoding worker; they run until ``close``."""
        if self._closed:
            raise RuntimeError("engine is closed")
        if self._sh is not None:
            return
        self._sh = sh = {"waiting": deque(), "ready": deque(), "finished": [], "lock": threading.Lock(), "prep_lock": threading.Lock(),
                         "jobs": queue.Queue(), "errors": [], "stop": False}

        def bulk_worker():
            with torch.inference_mode(), torch.cuda.stream(self.bulk_tts.model.bulk_stream):
                while True:
                    batch = sh["jobs"].get()
                    if batch is None:
                        return
                    try:
                        for a, wav in zip(batch, self._bulk(batch)):
                            a["future"].set_result(wav)
                    except BaseException as e:  # noqa: BLE001
                        sh["errors"].append(e)
                        for a in batch:
                            if not a["future"].done():
                                a["future"].set_exception(e)

        self._threads = [threading.Thread(target=bulk_worker, name="sopro-bulk", daemon=True)]
        self._threads += [threading.Thread(target=self._drive, args=(g, sh), name=f"sopro-gen{i}", daemon=True) for i, g in enumerate(self.gens)]
        for t in self._threads:
            t.start()
        atexit.register(self.close)  # worker threads must not be inside the HIP runtime when the interpreter tears it down

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.close()

    def _item(self, rq: Dict[str, Any]) -> Dict[str, Any]:
        cfg = self.model.cfg
        ids = rq["text_ids"] if rq.get("text_ids") is not None else self.tts.encode_text(rq["text"])
        if int(ids.numel()) == 0 or int(ids.numel()) > self.S_cap:
            raise ValueError(f"text of {int(ids.numel())} positions (this engine takes 1..{self.S_cap})")
        mf = min(int(rq.get("max_frames", self.max_frames)), self.max_frames)
        ss = float(rq["style_strength"] if rq.get("style_strength") is not None else cfg.style_strength)
        min_gen = int(rq["min_gen_frames"] if rq.get("min_gen_frames") is not None else cfg.min_gen_frames)
        prm = torch.tensor([float(rq.get("top_p", 0.9)), float(rq.get("temperature", 1.05)), 1.0 if rq.get("anti_loop", True) else 0.0,
                            0.85, 1.2, 1.1, 50.0, float(min_gen)], dtype=torch.float32)
        return {"ids": ids, "ref": rq["ref"], "mf": mf, "ss": ss, "prm": prm, "seed": rq.get("seed"), "future": Future()}

    def submit(self, **rq) -> "Future[torch.Tensor]":
        """Queue one utterance: text_ids | text, ref, max_frames, top_p, temperature, anti_loop, style_strength,
        min_gen_frames, seed.  The future resolves to the waveform [1, 1, N] on the device."""
        self.start()
        if self._sh["stop"]:
            raise RuntimeError("engine is closed")
        it = self._item(rq)
        with self._sh["lock"]:
            self._sh["waiting"].append(it)
        return it["future"]

    @torch.inference_mode()
    def run(self, requests: Sequence[Dict[str, Any]]) -> List[torch.Tensor]:
        """All requests at once (each the keyword dict of ``submit``); waveforms come back in request order."""
        self.start()
        sh = self._sh
        items = [self._item(rq) for rq in requests]
        with sh["lock"]:
            sh["waiting"].extend(items)
        self._prepare_some(sh, self.slots * len(self.gens))  # the initial fill in one batch
        out = [it["future"].result() for it in items]
        self.stats["utterances"] += len(requests)
        return out

    def _prepare_some(self, sh: Dict[str, Any], limit: int) -> None:
        """Move up to ``limit`` waiting requests (one style strength per group) to the ready list; one thread at a time."""
        if not sh["prep_lock"].acquire(blocking=False):
            return
        try:
            group: List[Dict[str, Any]] = []
            with sh["lock"]:
                w = sh["waiting"]
                while w and len(group) < limit and w[0]["ss"] == (group[0]["ss"] if group else w[0]["ss"]):
                    group.append(w.popleft())
            if group:
                try:
                    with torch.cuda.stream(self.model.prep_stream):  # not behind the frames the calling generator has queued
                        self._prepare(group)
                except BaseException as e:  # noqa: BLE001  (a bad request must not strand the others)
                    for it in group:
                        it["future"].set_exception(e)
                    return
                with sh["lock"]:
                    sh["ready"].extend(group)
        finally:
            sh["prep_lock"].release()

    def _drive(self, gen: Dict[str, Any], sh: Dict[str, Any]) -> None:
        """One slot set: admit / step / poll / harvest until no request is left for it."""
        import time

        plan, st = gen["plan"], gen["stream"]
        free = list(range(self.slots))[::-1]
        active: Dict[int, Dict[str, Any]] = {}
        gen["have_prev"], chunk = False, 0
        try:
            with torch.inference_mode(), torch.cuda.stream(st):
                while True:
                    with sh["lock"]:
                        take = [sh["ready"].popleft() for _ in range(min(len(free), len(sh["ready"])))]
                    for it in take:
                        row = free.pop()
                        self._admit(gen, row, it)
                        it["adm"] = chunk  # present in the snapshots of chunk `chunk` and later
                        active[row] = it
                    if not active:
                        with sh["lock"]:
                            idle = not sh["waiting"] and not sh["ready"]
                            if idle and sh["finished"]:  # nobody will fill that batch soon: send what there is
                                sh["jobs"].put(sh["finished"])
                                sh["finished"] = []
                        if idle and sh["stop"]:
                            return
                        if not idle:
                            self._prepare_some(sh, self.prep_batch)  # nothing to step: help with (or wait for) the preparation
                        time.sleep(0.0002 if not idle else 0.001)
                        continue
                    # ---- a chunk of frames for every slot, then a snapshot of the slot states behind it.  The host looks at
                    # the snapshot of the PREVIOUS chunk (the GPU never waits for the round trip) and prepares the next
                    # utterances while the frames run; a finished row idles one chunk more before it is harvested.
                    plan.steps(self.poll_every)
                    slot_i = gen["snap_i"] = (gen.get("snap_i", 0) + 1) & 1
                    if "snap" not in gen:
                        gen["snap"] = [hip.HostMirror(4 * self.slots) for _ in range(2)]
                        gen["snap_ev"] = [torch.cuda.Event() for _ in range(2)]
                    snap = torch.stack([plan.start, plan.stop_t, plan.first_eos, plan.ctr[0:1].expand(self.slots)]).contiguous()
                    gen["snap"][slot_i].copy_from(snap.view(-1))
                    gen["snap_ev"][slot_i].record(st)
                    self.stats["frames"] += self.poll_every
                    self.stats["slot_frames_used"] += self.poll_every * len(active)
                    if sh["waiting"] and len(sh["ready"]) < self.prep_batch:
                        self._prepare_some(sh, self.prep_batch)
                    with sh["lock"]:
                        more = bool(sh["waiting"] or sh["ready"])
                    prev = slot_i ^ 1
                    if gen["have_prev"] and more:
                        look, seen = prev, chunk - 1  # trailing poll: something may still want a slot, keep the GPU fed
                    else:
                        look, seen = slot_i, chunk    # nothing to admit any more: look at this chunk directly
                    gen["have_prev"] = True
                    chunk += 1
                    gen["snap_ev"][look].synchronize()
                    flat = gen["snap"][look].values()
                    start, stop_t, first_eos, stepv = (flat[i * self.slots:(i + 1) * self.slots] for i in range(4))
                    step = stepv[0]
                    # ---- harvest: EOS rule satisfied (model.py:301-305) or frame budget used up
                    done: List[Dict[str, Any]] = []
                    for row in list(active):
                        a = active[row]
                        if a["adm"] > seen:
                            continue  # admitted after that snapshot was taken: the slot's entries are its predecessor's
                        ran = step - start[row]
                        if stop_t[row] >= 0 or ran >= a["mf"] + 1:
                            T = first_eos[row] if first_eos[row] >= 0 else min(ran, a["mf"] + 1)  # cut at the FIRST EOS (model.py:385-390)
                            a["rvq1"] = plan.hist[row, :T].clone()
                            a["cond_t"] = plan.cond[row, :T].clone()
                            plan.start[row:row + 1].fill_(-1)
                            a["done"] = torch.cuda.Event()
                            a["done"].record(st)
                            a["T"] = T
                            done.append(a)
                            del active[row]
                            free.append(row)
                    if done:
                        with sh["lock"]:
                            sh["finished"].extend(done)
                            while len(sh["finished"]) >= self.bulk_batch:
                                sh["jobs"].put(sh["finished"][: self.bulk_batch])
                                sh["finished"] = sh["finished"][self.bulk_batch:]
        except BaseException as e:  # noqa: BLE001
            sh["errors"].append(e)
            with sh["lock"]:
                pend = list(active.values()) + list(sh["ready"]) + list(sh["waiting"]) + list(sh["finished"])
            for it in pend:
                if not it["future"].done():
                    it["future"].set_exception(e)

    def _bulk(self, batch: List[Dict[str, Any]]) -> List[torch.Tensor]:
        bt = self.bulk_tts
        m, dev = bt.model, bt.model.device
        lens = [int(a["T"]) for a in batch]
        B, Tm = len(batch), max(lens)
        hop = int(bt.codec.mc.frame_samples)
        if Tm == 0:
            return [torch.zeros(1, 1, 0, device=dev) for _ in batch]
        Tm = -(-Tm // 8) * 8
        cur = torch.cuda.current_stream(dev)
        cond = torch.zeros(B, Tm, m.D, device=dev)
        rvq1 = torch.zeros(B, Tm, dtype=torch.int32, device=dev)
        for b, a in enumerate(batch):
            cur.wait_event(a["done"])
            a["cond_t"].record_stream(cur)  # made on the generation stream, read here
            a["rvq1"].record_stream(cur)
            cond[b, : lens[b]] = a["cond_t"]
            rvq1[b, : lens[b]] = a["rvq1"].clamp(max=m.V - 1)
        toks = m.nar_refine(cond, rvq1, lens=[max(1, n) for n in lens])
        codes = torch.zeros(B, Tm, m.Q, dtype=torch.long, device=dev)
        for b in range(B):
            codes[b, : lens[b]] = toks[b, : lens[b]]
        wav = bt.codec.decode_batch(codes)
        self.stats["bulk_batches"] += 1
        return [wav[b, : lens[b] * hop].reshape(1, 1, -1) for b in range(B)]
