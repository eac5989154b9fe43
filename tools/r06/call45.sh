#!/bin/bash
# r06 call 45: whole suite (with the tile-shape test of the fused-RMSNorm forms) on the final product library, then the profile half of the evidence set
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c45; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " > $O/tile_life_final.txt
bash tools/collect_evidence.sh r06 profiles > $O/profiles.log 2>&1; tail -3 $O/profiles.log
