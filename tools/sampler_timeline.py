#!/usr/bin/env python
"""Developer probe: shader-clock stamps inside ar_sample_kernel (sopro_ar_state.dbg): where the sampler's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
B, Tar, D, V = 32, 64, 384, 2048
z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
x, cond, emb = z(B, D), torch.randn(B, Tar, D, device=DEV), torch.randn(2 * V + 1, D, device=DEV)
hist, ctr, feos, stop, prm, rec, nonce = z(B, Tar, dt=torch.int32), z(8, dt=torch.int32), z(B, dt=torch.int32), z(B, dt=torch.int32), z(8), z(B, 64, dt=torch.int32), z(B, dt=torch.int32)
dbg = z(B, 12, dt=torch.int64)
st = hip.ArState()
st.x_cur, st.cond, st.emb, st.hist = x.data_ptr(), cond.data_ptr(), emb.data_ptr(), hist.data_ptr()
row_step = z(B, dt=torch.int32)
st.step, st.row_step, st.n_stopped = ctr.data_ptr(), row_step.data_ptr(), ctr.data_ptr() + 8
st.first_eos, st.stop_t, st.params, st.recent, st.nonce, st.dbg = feos.data_ptr(), stop.data_ptr(), prm.data_ptr(), rec.data_ptr(), nonce.data_ptr(), dbg.data_ptr()
st.seed, st.B, st.D, st.Tar, st.max_steps, st.V, st.bos_row = 7, B, D, Tar, Tar, V, 2 * V
for name, tp in (("sampling top_p=0.9", 0.9), ("greedy", 0.0)):
    prm.copy_(torch.tensor([tp, 1.05, 1.0, 0.85, 1.2, 1.1, 50.0, 12.0]))
    hip.ar_init(st)
    lg = torch.randn(B, V + 1, device=DEV) * 2
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for t in range(40):
        if t == 30:
            ev[0].record()
        hip.ar_sample(st, lg, V + 1)
    ev[1].record()
    torch.cuda.synchronize()
    d = dbg.cpu().double()
    names = ["start", "loads + policy", "nan/temp+penalty", "max/thr+barrier", "compaction+barrier", "all-pairs+barrier", "tail (token)", "next input stores", "recent/hist/counters"]
    d[:, 9] = d[:, 8]
    d[:, 10] = torch.maximum(d[:, 10], d[:, 8])
    seg = [float((d[:, i + 1] - d[:, i]).median()) for i in range(8)]
    print(f"{name}: {ev[0].elapsed_time(ev[1]) / 10 * 1e3:.1f} us per launch (events, back to back); median cycles per phase: " +
          ", ".join(f"{n} {s:.0f}" for n, s in zip(names[1:], seg)) + f"; total {float((d[:, 10] - d[:, 0]).median()):.0f} cyc; start spread {float(d[:, 0].max() - d[:, 0].min()):.0f}")
