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

import threading
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from . import hip


class PipelinedSynthesizer:
    def __init__(self, tts, lanes: int = 2, ar_cus: int = 64):
        self.device = tts.device
        total = hip.device_info(self.device.index or 0)["cus"]
        if not (0 < ar_cus < total):
            raise ValueError("ar_cus must leave CUs for the bulk phase")
        self.lanes = []
        self._saved = (tts.model.stream, tts.model.bulk_stream, tts.codec.stream)
        self._streams = []
        for i in range(int(lanes)):
            lane = tts if i == 0 else tts.clone_lane()
            lane.model.stream = hip.cu_range_stream(0, ar_cus, self.device)
            lane.model.bulk_stream = hip.cu_range_stream(ar_cus, total - ar_cus, self.device)
            self._streams += [lane.model.stream, lane.model.bulk_stream]
            lane.codec.stream = lane.model.bulk_stream
            lane.model._ar_cache.clear()  # recorded graphs belong to the stream they were captured on
            self.lanes.append(lane)
        self.ar_lock, self.bulk_lock = threading.Lock(), threading.Lock()
        self.ar_cus, self.bulk_cus = ar_cus, total - ar_cus

    def close(self) -> None:
        """Drop the extra lanes, destroy the CU-masked streams (and the graphs recorded on them) and give lane 0
        (the caller's engine) its full-chip streams back."""
        if not self.lanes:
            return
        torch.cuda.synchronize(self.device)
        for lane in self.lanes:
            lane.model._ar_cache.clear()  # hipGraphExecDestroy now, not at interpreter shutdown
            lane.model.ws.clear()
            lane.codec.ws.clear()
        lane0 = self.lanes[0]
        lane0.model.stream, lane0.model.bulk_stream, lane0.codec.stream = self._saved
        self.lanes = []
        torch.cuda.synchronize(self.device)
        for st in self._streams:
            hip.destroy_stream(st)
        self._streams = []

    def run(self, jobs: Sequence[Dict[str, Any]], timings: Optional[Dict[str, float]] = None) -> List[Any]:
        """Each job is the keyword dict of ``SoproTTS.synthesize_batch``; results come back in job order."""
        results: List[Any] = [None] * len(jobs)
        errors: List[BaseException] = []
        nxt = [0]
        pick = threading.Lock()

        def worker(lane):
            # the worker's current stream is the lane's own (never the NULL stream, which would serialise the lanes)
            with torch.cuda.stream(lane.model.stream):
                while True:
                    with pick:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(jobs) or errors:
                        return
                    try:
                        results[i] = lane.synthesize_batch(phase_locks=(self.ar_lock, self.bulk_lock), timings=timings, **jobs[i])
                    except BaseException as e:  # noqa: BLE001
                        errors.append(e)
                        return

        threads = [threading.Thread(target=worker, args=(lane,)) for lane in self.lanes[: max(1, min(len(self.lanes), len(jobs)))]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results
