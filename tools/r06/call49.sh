#!/bin/bash
# r06 call 49: tile lives of the refinement's GLU and stage-head launches (developer library: epilogue stamps)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c49; mkdir -p $O; cd $R
SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so timeout 600 python tools/r06/tile_life.py 2>&1 | grep -v amdgpu | tee $O/tile_life.txt | cut -c1-44,120-400 | head -12
