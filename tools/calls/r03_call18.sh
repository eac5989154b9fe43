#!/bin/bash
# round 3, call 18: fragment reads ahead of the MFMAs in the SEANet kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03j; O=gpurun_out/r03j
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "seanet or tail or res128 or up128 or codec or mimi" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" ; tail -3 $O/pytest_a.log
PROBE_B=32 timeout 300 python tools/mimi_probe.py 2>&1 | grep -v "amdgpu.ids\|Exception ignored\|Traceback\|hip.py\|AttributeError" | tee $O/mimi_probe.txt
