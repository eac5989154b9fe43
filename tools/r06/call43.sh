#!/bin/bash
# r06 call 43: fused-RMSNorm + GLU form, previous against new library: how many elements differ, by how much, in which columns
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_prev.so python tools/r06/form_dump.py /tmp/prev.pt 2>&1 | grep -v amdgpu
python tools/r06/form_dump.py /tmp/new.pt 2>&1 | grep -v amdgpu
python - <<PY
import torch
a, b = torch.load("/tmp/prev.pt"), torch.load("/tmp/new.pt")
for k in ("pre", "glu"):
    x, y = a[k], b[k]
    d = (x != y)
    print(k, "differing elements", int(d.sum()), "of", x.numel(), "max abs diff", float((x - y).abs().max()), "max |x|", float(x.abs().max()))
    if d.any():
        cols = d.any(0).nonzero().flatten()
        print("  columns with differences:", cols[:40].tolist(), "... count", len(cols))
        rows = d.any(1).nonzero().flatten()
        print("  rows with differences: count", len(rows), rows[:20].tolist())
PY
