import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_engine, make_inputs
tts, cfg, mc, wn, mn = build_engine("cuda:0")
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
m = tts.model
prep = m.prepare_conditioning_batch(ids[:1], [ref], max_frames=199)
tok0 = torch.randint(0, 2048, (1, 200), device="cuda:0")
def run(T, graph=True):
    m.use_graph = graph
    o = m.nar_refine(prep["cond_ar"][:, :T], tok0[:, :T]).cpu()
    m.use_graph = True
    return o
want = {T: run(T, False) for T in (200, 6, 12, 48)}
a = run(200); b = run(200)           # eager, record
print("200 after record:", int((a != want[200]).sum()), int((b != want[200]).sum()))
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "many"):
    for T in range(1, 49):
        run(T)
    print("200 after 48 eager shapes:", int((run(200) != want[200]).sum()), flush=True)
if mode in ("all", "rec"):
    for T in (6, 12, 18, 24, 30, 36, 42, 48):
        run(T)                       # second sight: record
    print("200 after 8 recordings:", int((run(200) != want[200]).sum()), "| 6:", int((run(6) != want[6]).sum()), "12:", int((run(12) != want[12]).sum()), "48:", int((run(48) != want[48]).sum()), flush=True)
if mode in ("all", "big"):
    p32 = m.prepare_conditioning_batch(ids, [ref] * 32, max_frames=399)
    t32 = torch.randint(0, 2048, (32, 400), device="cuda:0")
    m.nar_refine(p32["cond_ar"][:, :400], t32)
    print("200 after a 32x400 batch:", int((run(200) != want[200]).sum()), flush=True)
    tts.codec.decode_batch(torch.randint(0, 2048, (32, 400, 32), device="cuda:0"))
    tts.codec.decode_batch(torch.randint(0, 2048, (32, 200, 32), device="cuda:0"))
    print("200 after big decodes:", int((run(200) != want[200]).sum()), "ws bytes", m.ws.bytes >> 20, tts.codec.ws.bytes >> 20, flush=True)
if mode == "diag":
    for T in range(1, 9):
        run(T)
    o1 = run(200); o2 = run(200)
    d1 = (o1 != want[200])[0]
    print("replay after eager:", int(d1.sum()), "per-codebook mismatches:", d1.sum(0).tolist())
    print("rows with mismatches:", d1.any(1).nonzero().flatten().tolist()[:20], "...", int(d1.any(1).sum()))
    print("second replay:", int((o2 != want[200]).sum()))
    ws = m.ws._bufs
    for k, v in ws.items():
        if k[0] in ("nar.lens", "nar.rvq1") and (k[1] == (1,) or k[1] == (1, 200)):
            print(k, v.flatten()[:8].tolist())
    c = ws[("nar.cond", (200, 384), torch.float32)]
    print("cond buffer equals input:", bool(torch.equal(c.view(1, 200, 384), prep["cond_ar"][:, :200].float())))
