#!/bin/bash
# r06 call 32: where a tile of the tile kernel spends its life (shader-clock stamps), product library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c32; mkdir -p $O; cd $R
timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life.txt | cut -c1-400
