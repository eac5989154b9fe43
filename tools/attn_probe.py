#!/usr/bin/env python
"""Developer probe (round 3): the reference cross-attention of the conditioning (B x 2 heads x 192, Tq = 201 queries, Tk = 150
keys) and the codec's window attention (8 heads x 64, 400 positions) on the matrix cores against the LDS / VALU kernel
(SOPRO_ATTN_VALU=1 selects it), alone on the whole chip.  Prints us per launch and TFLOP/s.
    python tools/attn_probe.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch

    from sopro_amd import hip

    dev = torch.device("cuda:0")
    for name, B, H, dh, Tq, Tk, causal in (("ref xattn 64 rows", 64, 2, 192, 201, 150, False), ("ref xattn 32 rows", 32, 2, 192, 201, 150, False),
                                           ("codec window 64 rows", 64, 8, 64, 400, 400, True)):
        D = H * dh
        q, k, v = (torch.randn(B, t, D, device=dev) for t in (Tq, Tk, Tk))
        o = torch.empty(B, Tq, D, device=dev)
        kl = torch.full((B,), Tk, dtype=torch.int32, device=dev)
        kw = dict(B=B, H=H, dh=dh, Tq=Tq, Tk=Tk, ldq=D, ldk=D, ldv=D, ldo=D, q_bstride=Tq * D, k_bstride=Tk * D, v_bstride=Tk * D, o_bstride=Tq * D,
                  klens=None if causal else kl, causal=causal, window=250 if causal else 0)
        for _ in range(3):
            hip.attention(q, k, v, o, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.attention(q, k, v, o, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        pairs = Tq * Tk if not causal else sum(min(t + 1, 250) for t in range(Tq))
        print(f"{'valu' if os.environ.get('SOPRO_ATTN_VALU') else 'mfma'}: {name:22s} {us:8.1f} us  {4.0 * dh * pairs * B * H / us / 1e6:6.1f} TFLOP/s", flush=True)
else:
    for valu in ("0", "1"):
        env = dict(os.environ)
        env.pop("SOPRO_ATTN_VALU", None)
        if valu == "1":
            env["SOPRO_ATTN_VALU"] = "1"  # (the library tests the variable's presence)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
