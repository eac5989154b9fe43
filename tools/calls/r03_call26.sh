#!/bin/bash
# round 3, call 26: the sixteen-wave tail kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03r; O=gpurun_out/r03r
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tail" 2>&1 | tail -5
python tools/tail_probe.py 32 2>&1 | grep -v amdgpu
python tools/tail_probe.py 64 2>&1 | grep -v amdgpu
python tools/tail_probe.py 32 384000 16 2>&1 | grep -v amdgpu
PROBE_B=32 timeout 200 python tools/mimi_probe.py 2>&1 | grep -v "amdgpu.ids\|Exception\|Traceback\|hip.py\|Attribute" | tee $O/mimi.txt
