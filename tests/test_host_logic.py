"""Host-side decisions that need no GPU: the split-K heuristic, the scratch budget, the style of batching keys."""
import torch

from sopro_amd import hip
from sopro_amd.model import Workspace


def test_auto_ksplit_only_for_few_tiles_and_long_k():
    # whole batches fill the chip: never split
    assert hip._auto_ksplit(12800, 1536, 512, 2, hip.EPI_NONE) == 1
    assert hip._auto_ksplit(6400, 384, 1536, 3, hip.EPI_RES) == 1
    # a streaming chunk: a handful of 64x64 tiles; split from K = 1024 up, not below (the release / acquire pair costs ~10 us)
    assert hip._auto_ksplit(6, 384, 1536, 3, hip.EPI_RES) > 1
    assert hip._auto_ksplit(12, 512, 2048, 2, hip.EPI_RES) > 1
    assert hip._auto_ksplit(6, 768, 384, 3, hip.EPI_GLU) == 1
    assert hip._auto_ksplit(12, 2048, 256, 3, hip.EPI_NONE) == 1
    for (M, N, K, pieces, epi) in [(6, 384, 1536, 3, hip.EPI_RES), (12, 4096, 2048, 2, hip.EPI_NONE), (33, 4096, 2048, 2, hip.EPI_NONE),
                                   (1, 64, 4096, 3, hip.EPI_GELU)]:
        ks = hip._auto_ksplit(M, N, K, pieces, epi)
        bm, bn = (64, 128) if (pieces == 3 and epi == hip.EPI_GLU) else ((64, 64) if (pieces == 3 or N <= 64 or M <= 64) else (128, 128))
        tiles = -(-M // bm) * -(-N // bn)
        assert 1 <= ks <= 16 and ks * tiles * bm * bn * 4 <= hip._SPLITK_WS_BYTES and tiles <= hip._SPLITK_TICKETS


def test_workspace_keeps_one_buffer_per_shape_and_counts_bytes():
    ws = Workspace(torch.device("cpu"))
    a = ws.get("x", (4, 8))
    assert ws.get("x", (4, 8)) is a and ws.bytes == 4 * 8 * 4
    b = ws.get("x", (2, 8), zero=True)
    assert b is not a and float(b.abs().sum()) == 0.0 and ws.bytes == (32 + 16) * 4
    c = ws.get("i", (3,), dtype=torch.int32)
    assert c.dtype == torch.int32 and ws.bytes == (32 + 16 + 3) * 4
    assert ws.over(100) and not ws.over(1000)
    ws.clear()
    assert ws.bytes == 0 and ws.get("x", (4, 8)) is not a


def test_workspace_tells_every_sharer_when_its_buffers_go():
    """Scratch shared between engines (MimiCodec.share_scratch: the lanes of a pipeline decode in one set of buffers): dropping the buffers must
    drop every sharer's recorded launch sequences first - they hold raw pointers into them."""
    ws = Workspace(torch.device("cpu"))
    dropped = []
    ws.on_clear.append(lambda: dropped.append("lane0"))
    ws.on_clear.append(lambda: dropped.append("lane1"))
    ws.get("x", (8,))
    ws.clear()
    assert dropped == ["lane0", "lane1"] and ws.bytes == 0
    ws.clear()
    assert dropped == ["lane0", "lane1"] * 2  # (idempotent for the listeners)


def test_no_garbage_collection_while_a_launch_sequence_is_recorded():
    """hip.capture_begin pauses the cyclic collector until capture_end (a finalizer that frees device memory on the recording thread
    invalidates the recording); nested / repeated pauses restore the state they found."""
    import gc

    was = gc.isenabled()
    try:
        gc.enable()
        hip._gc_pause()
        assert not gc.isenabled()
        hip._gc_pause()
        hip._gc_resume()
        assert not gc.isenabled()
        hip._gc_resume()
        assert gc.isenabled()
        gc.disable()
        hip._gc_pause()
        hip._gc_resume()
        assert not gc.isenabled()  # it was off before: stays off
    finally:
        gc.enable() if was else gc.disable()


def test_bench_line_stays_under_the_drivers_parse_buffer():
    """Round 5's 21 kB line came back from the driver as parsed = null.  bench.compact_line() keeps the headline, config (second
    metric, legs), roofline (+ batch 1), roofline_more, cpu_baseline and parity of that very record under 8 kB."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r05_bench_line.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT == 8192
    assert line["value"] == full["value"] and line["metric"] == full["metric"] and line["dtype"] == "f32"
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["roofline"]["batch1"]["frac"] == full["roofline"]["batch1"]["frac"]
    assert [e["frac"] for e in line["roofline_more"]] == [e["frac"] for e in full["roofline_more"]]
    assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and line["cpu_baseline"]["kind"] == "port"
    assert line["config"]["second_metric"]["ttfa_ms_p50"] == full["config"]["second_metric"]["ttfa_ms_p50"]
    assert set(line["config"]["legs"]) == set(full["config"]["legs"]) and line["parity"]["ok"] is True
    assert "workload" in line["config"] and not any(k in line["config"] for k in ("model", "seq_len"))
