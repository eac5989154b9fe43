#!/bin/bash
# r06 call 28: tile kernel on 1 x 4 waves of 128 x 32 (no W fragment requested twice per workgroup) against the product's 2 x 2 waves of 64 x 64
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c28; mkdir -p $O; cd $R
SOPRO_DEV=1 timeout 900 python tools/r06/tile_probe.py 1 3 2>&1 | grep -v amdgpu.ids | tee $O/tile_probe.txt | cut -c1-260
