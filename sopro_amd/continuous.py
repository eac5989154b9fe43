"""Frame-level admission into the batched AR graph ("slot-based admission").

``synthesize_batch`` generates a batch until its LAST utterance has finished: with ragged lengths the rows that reached
end-of-speech idle for the rest of the batch.  Here the AR frame graph runs over a fixed set of slots; every slot has its own
time base on the device (``sopro_ar_state.start / row_max / row_params``), a slot whose utterance has finished is
harvested at the next poll and handed a queued utterance (conditioning rows, folded text operands, zeroed ring columns,
``sopro_ar_admit``), and finished utterances are refined and decoded in batches of their own.  Per-utterance results are
the ones ``synthesize`` gives (greedy decode: identical tokens): a slot's arithmetic never depends on its neighbours.
"""
from __future__ import annotations

from collections import deque
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import hip
from .model import PreparedReference, _ARPlan


class ContinuousSynthesizer:
    def __init__(self, tts, *, slots: int = 32, max_frames: int = 400, max_text: int = 128, poll_every: int = 16,
                 bulk_batch: int = 16):
        self.tts, self.model, self.codec = tts, tts.model, tts.codec
        self.slots, self.poll_every, self.bulk_batch = int(slots), int(poll_every), int(bulk_batch)
        self.max_frames = int(max_frames)
        m = self.model
        s_cap = ((int(max_text) + 63) // 64) * 64
        with m.on_stream():
            self.plan = _ARPlan(m, self.slots, s_cap, self.max_frames + 1, slots=True)
            hip.ar_init(self.plan.state)
        self.plan.ensure_graph()
        self.stats = {"frames": 0, "slot_frames_used": 0, "utterances": 0, "bulk_batches": 0}

    @torch.inference_mode()
    def run(self, requests: Sequence[Dict[str, Any]]) -> List[torch.Tensor]:
        """Each request: dict(text_ids | text, ref, max_frames=…, top_p=…, temperature=…, anti_loop=…, style_strength=…,
        min_gen_frames=…).  Returns the waveforms [1, 1, N] in request order."""
        m, plan, cfg = self.model, self.plan, self.model.cfg
        dev = m.device
        queue = deque(enumerate(requests))
        free = list(range(self.slots))[::-1]
        active: Dict[int, Dict[str, Any]] = {}
        finished: List[Dict[str, Any]] = []
        results: List[Optional[torch.Tensor]] = [None] * len(requests)
        while queue or active:
            # ---- admission: fill the free slots (conditioning per utterance, then launches on the AR stream)
            while free and queue:
                idx, rq = queue.popleft()
                ids = rq["text_ids"] if rq.get("text_ids") is not None else self.tts.encode_text(rq["text"])
                mf = min(int(rq.get("max_frames", self.max_frames)), self.max_frames)
                ss = float(rq["style_strength"] if rq.get("style_strength") is not None else cfg.style_strength)
                prep = m.prepare_conditioning(ids, rq["ref"], max_frames=mf, style_strength=ss)
                row = free.pop()
                min_gen = int(rq["min_gen_frames"] if rq.get("min_gen_frames") is not None else cfg.min_gen_frames)
                prm = torch.tensor([float(rq.get("top_p", 0.9)), float(rq.get("temperature", 1.05)), 1.0 if rq.get("anti_loop", True) else 0.0,
                                    0.85, 1.2, 1.1, 50.0, float(min_gen)], dtype=torch.float32)
                with m.on_stream():
                    plan.load_row(row, prep["cond_ar"][0], prep["txt_seq"][0])
                    plan.row_max[row:row + 1].fill_(mf + 1)
                    plan.row_params[row].copy_(prm, non_blocking=False)
                    hip.ar_admit(plan.state, row)
                active[row] = {"idx": idx, "budget": mf + 1}
            # ---- one chunk of frames for every slot
            with torch.cuda.stream(m.stream):
                for _ in range(self.poll_every):
                    plan.step()
                step = int(plan.ctr[0].item())  # the poll: device -> host once per chunk
                start = plan.start.tolist()
                stop_t = plan.stop_t.tolist()
                first_eos = plan.first_eos.tolist()
            self.stats["frames"] += self.poll_every
            self.stats["slot_frames_used"] += self.poll_every * len(active)
            # ---- harvest: EOS rule satisfied (model.py:301-305) or frame budget used up
            for row in list(active):
                a = active[row]
                ran = step - start[row]
                if stop_t[row] >= 0 or ran >= a["budget"]:
                    T = first_eos[row] if first_eos[row] >= 0 else min(ran, a["budget"])  # cut at the FIRST EOS (model.py:385-390)
                    with torch.cuda.stream(m.stream):
                        a["rvq1"] = plan.hist[row, :T].clone()
                        a["cond"] = plan.cond[row, :T].clone()
                        plan.start[row:row + 1].fill_(-1)
                    a["T"] = T
                    finished.append(a)
                    del active[row]
                    free.append(row)
            # ---- refinement + decoding of finished utterances, in batches of their own
            while len(finished) >= self.bulk_batch or (finished and not queue and not active):
                batch, finished = finished[: self.bulk_batch], finished[self.bulk_batch:]
                for a, wav in zip(batch, self._bulk(batch)):
                    results[a["idx"]] = wav
        self.stats["utterances"] += len(requests)
        return results  # type: ignore[return-value]

    def _bulk(self, batch: List[Dict[str, Any]]) -> List[torch.Tensor]:
        m, dev = self.model, self.model.device
        lens = [int(a["T"]) for a in batch]
        B, Tm = len(batch), max(lens)
        hop = int(self.codec.mc.frame_samples)
        if Tm == 0:
            return [torch.zeros(1, 1, 0, device=dev) for _ in batch]
        Tm = -(-Tm // 8) * 8
        cond = torch.zeros(B, Tm, m.D, device=dev)
        rvq1 = torch.zeros(B, Tm, dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).wait_stream(m.stream)
        for b, a in enumerate(batch):
            cond[b, : lens[b]] = a["cond"]
            rvq1[b, : lens[b]] = a["rvq1"].clamp(max=m.V - 1)
        toks = m.nar_refine(cond, rvq1, lens=[max(1, n) for n in lens])
        codes = torch.zeros(B, Tm, m.Q, dtype=torch.long, device=dev)
        for b in range(B):
            codes[b, : lens[b]] = toks[b, : lens[b]]
        wav = self.codec.decode_batch(codes)
        self.stats["bulk_batches"] += 1
        return [wav[b, : lens[b] * hop].reshape(1, 1, -1) for b in range(B)]
