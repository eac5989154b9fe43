#!/bin/bash
# r06 call 34: stamps inside the epilogue (developer library): transposition / first batch of rows / second batch
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c34; mkdir -p $O; cd $R
SOPRO_DEV=1 timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_epi.txt | cut -c1-460
