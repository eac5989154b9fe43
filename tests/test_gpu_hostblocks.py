"""Page-locked parameter blocks of the engine (SoproTTSModel._host_block) and the recorded launch sequences that hold their
addresses: a serving process walks through many (batch, text length) shapes; the refinement's recorded sequences must
keep reading live lengths and writing a live range word whatever the LRU of the other blocks does.  Expected tokens: the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import VOCAB

pytestmark = pytest.mark.gpu


def test_recorded_refinement_survives_the_walk_through_the_block_lru(tts, cfg, w):
    from oracle import sopro_oracle as O

    m = tts.model
    rng = np.random.default_rng(5)
    ref_tq = torch.from_numpy(rng.integers(0, 2048, size=(20, 32)))
    ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
    oref = O.prepare_reference(ref_tq, w, cfg)
    kw = dict(max_frames=7, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.0)
    ids0 = torch.from_numpy(rng.integers(1, VOCAB, size=9))
    want = O.generate_tokens(ids0, oref, w, cfg, **kw)

    def run():
        got = m.generate_tokens(ids0, ref, **kw)
        assert torch.equal(got.cpu(), want)

    run(); run(); run()  # eager, recorded, replayed
    lens_blk, range_blk = m._recorded_blocks[("nar.lens", 1)], m._recorded_blocks[("nar.range",)]
    p_lens, p_range = lens_blk.ptr, range_blk.ptr
    n_graphs = len(m._nar_graphs.graphs)
    assert n_graphs >= 1
    # walk through > 64 parameter-block keys: text lengths over several 32-token capacity classes x batch sizes, plus direct keys
    for S in (3, 33, 65, 97, 129):
        for B in (1, 2, 3):
            ids = [torch.from_numpy(rng.integers(1, VOCAB, size=S)) for _ in range(B)]
            m.prepare_conditioning_batch(ids, [ref] * B, max_frames=3, style_strength=1.0)
    for i in range(80):
        hb = m._host_block(("test.walk", i), 4 + (i % 3))
        hb.array()[:] = i
    assert len(m._host_blocks) <= 64
    # a key can be inherited by a user of another size (keys carry id(plan); a collected plan's id is re-used): the block is exact
    assert m._host_block(("test.size",), 3).n == 3 and m._host_block(("test.size",), 1).n == 1 and m._host_block(("test.size",), 4).n == 4
    assert ("test.walk", 0) not in m._host_blocks and ("test.walk", 79) in m._host_blocks
    # the blocks the recorded sequences hold are where they were, and the replay still gives the oracle's tokens
    assert m._recorded_blocks[("nar.lens", 1)].ptr == p_lens and m._recorded_blocks[("nar.range",)].ptr == p_range
    assert len(m._nar_graphs.graphs) >= n_graphs
    run()
    # text lengths inside one capacity class share a block; the LRU keeps the most recently used
    k0 = len(m._host_blocks)
    for S in (34, 40, 64):
        m.prepare_conditioning_batch([torch.from_numpy(rng.integers(1, VOCAB, size=S))], [ref], max_frames=3, style_strength=1.0)
    assert len(m._host_blocks) <= max(k0, 64) and ("cond.in", 1, 64) in m._host_blocks
    # dropping the recorded sequences drops their blocks with them, and the next passes record again
    m._drop_recorded()
    assert not m._recorded_blocks and not m._nar_graphs.graphs and lens_blk.ptr is None
    run(); run(); run()
    torch.cuda.synchronize()


def test_decode_parts_refuses_more_rows_than_its_workspace_holds(tts):
    """sopro_mimi_workspace_bytes sizes ONE chunk: the parts API must refuse B above sopro_mimi_chunk_rows(B, T) instead of
    running past the workspace; sopro_mimi_decode chunks by itself."""
    from sopro_amd import hip

    lib, eng = hip.load(), tts.codec.eng
    B, T = 40, 400
    rows = int(lib.sopro_mimi_chunk_rows(B, T))
    assert 0 < rows < B
    nbytes = int(lib.sopro_mimi_workspace_bytes(eng.h, B, T))
    assert nbytes == int(lib.sopro_mimi_workspace_bytes(eng.h, rows, T))
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda:0")  # never touched: the call is refused before any launch
    rc = lib.sopro_mimi_decode_parts(eng.h, ws.data_ptr(), ws.data_ptr(), B, T, ws.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "chunk" in lib.sopro_last_error().decode()
    torch.cuda.synchronize()
