#!/bin/bash
# r06 call 33: EPI_RES with all residual pieces requested in front of the epilogue's transposition (developer library) against the
# product library (requested batch by batch inside the store loop): tile lives, per-shape times, refinement pass, decode, tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c33; mkdir -p $O; cd $R
echo "--- batch by batch"; timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_old.txt | cut -c1-330
echo "--- ahead"; SOPRO_DEV=1 timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_new.txt | cut -c1-330
for i in 1 2; do
  echo "batch by batch:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  echo "ahead:"; SOPRO_DEV=1 timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; SOPRO_DEV=1 timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
done
SOPRO_DEV=1 timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest (ahead) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_new.log | cut -c1-260 | tail -12
