#!/bin/bash
# r06 call 16: cost of the GELU epilogue / fused-RMSNorm staging in the refinement's contractions
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c16; mkdir -p $O; cd $R
timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/ff_cost.txt | cut -c1-220
