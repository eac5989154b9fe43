#!/bin/bash
# r06 call 35: calls 28 / 30 / 31 / 33 compared the product library with ITSELF (SOPRO_DEV=1 switches the Python side's developer environment on; the
# developer LIBRARY is chosen with SOPRO_HIP_LIB).  Again, properly: A = product library (round's evidence library), B = developer library
# (EPI_RES residual pieces requested ahead + the 1 x 4-wave tile as override 3), F = B + the staging folded into the MFMA stream.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c35; mkdir -p $O; cd $R
B="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"; F="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_fold.so"
echo "--- A tile life"; timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_A.txt | cut -c60-330
echo "--- B tile life"; env $B timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_B.txt | cut -c60-460
[ -f sopro_amd/libsopro_hip_fold.so ] && { echo "--- F tile life"; env $F timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_F.txt | cut -c60-460; }
echo "--- B tiles 1 / 3"; env $B timeout 600 python tools/r06/tile_probe.py 1 3 2>&1 | grep " x " | tee $O/tile_probe_B.txt | cut -c1-200
for i in 1 2; do
  echo "A:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  echo "B:"; env $B timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; env $B timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  [ -f sopro_amd/libsopro_hip_fold.so ] && { echo "F:"; env $F timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; env $F timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"; }
done
