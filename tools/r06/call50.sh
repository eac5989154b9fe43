#!/bin/bash
# r06 call 50: GLU epilogue with every lane working (developer library) against the product library: bits of every form, refinement pass x 3,
# tile life of the GLU launch, whole suite on the developer library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c50; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/prod.txt
env $D timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/dev.txt
paste -d'|' $O/prod.txt $O/dev.txt | awk -F'|' '{split($1,a,": "); split($2,b,": "); print a[1] ": " a[2] " " b[2] (a[2]==b[2] ? "" : "   <-- differs")}' | tee $O/form_hash.txt | grep -c differs
for i in 1 2 3; do
  echo "product:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "every lane:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
env $D timeout 600 python tools/r06/tile_life.py 2>&1 | grep "glu" | cut -c1-44,120-400
env $D timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_dev.log 2>&1; echo "pytest gpu (developer library) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_dev.log | cut -c1-260 | tail -8
