#!/bin/bash
# r06 call 48: the 1 x 4-wave tile (developer override 3) on the final epilogue: bits and time against the 2 x 2 form, twice
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c48; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
for i in 1 2; do env $D timeout 600 python tools/r06/tile_probe.py 1 3 2>&1 | grep " x " | tee -a $O/tile_probe.txt | cut -c1-200; done
for i in 1 2; do
  echo "2x2:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "1x4 (f16 family):"; env $D SOPRO_F16X3_TILE=3 timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
