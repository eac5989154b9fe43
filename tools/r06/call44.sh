#!/bin/bash
# r06 call 44: do the fused-RMSNorm forms' bits depend on the tile shape - previous library, new library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
echo "prev:"; SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_prev.so python tools/r06/form_tiles.py 2>&1 | grep " x "
echo "new:"; python tools/r06/form_tiles.py 2>&1 | grep " x "
