#!/bin/bash
# r06 call 39: epilogue with a straight-line path for whole tiles inside one segment (developer library) against the product library
# (= the library of call 38): tile lives, refinement pass, decode, whole suite on the developer library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c39; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
echo "--- product"; timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_A.txt | cut -c1-44,120-280
echo "--- straight-line"; env $D timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_B.txt | cut -c1-44,120-400
for i in 1 2; do
  echo "A:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  echo "B:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; env $D timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
done
env $D timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_dev.log 2>&1; echo "pytest gpu (developer library) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_dev.log | cut -c1-260 | tail -8
