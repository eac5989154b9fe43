#!/bin/bash
# r06 call 31: tile kernel with the next step's staging placed inside the MFMA stream (FOLD, developer library) against the product library
# (separate lstore, same source): per-shape times, refinement pass, decode, tests on the folded library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c31; mkdir -p $O; cd $R
for i in 1 2; do
  echo "--- separate lstore (run $i)"; timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep " x " | tee -a $O/ff_sep.txt | cut -c1-200
  echo "--- folded (run $i)"; SOPRO_DEV=1 timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep " x " | tee -a $O/ff_fold.txt | cut -c1-200
done
echo "--- separate"; timeout 600 python tools/r06/tile_probe.py 1 2>&1 | grep " x " | tee $O/tile_sep.txt | cut -c1-200
echo "--- folded"; SOPRO_DEV=1 timeout 600 python tools/r06/tile_probe.py 1 2>&1 | grep " x " | tee $O/tile_fold.txt | cut -c1-200
for i in 1 2; do
  echo "separate:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  echo "folded:"; SOPRO_DEV=1 timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; SOPRO_DEV=1 timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
done
SOPRO_DEV=1 timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_fold.log 2>&1; echo "pytest (folded) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_fold.log | cut -c1-260 | tail -12
