#!/bin/bash
# r06 call 54: arg-max epilogue scan by 16-byte LDS reads (product library) against the element walk (developer library of call 52): refinement
# pass x 3, form hashes, whole suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c54; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/prod.txt
env $D timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/dev.txt
paste -d'|' $O/dev.txt $O/prod.txt | awk -F'|' '{split($1,a,": "); split($2,b,": "); print a[1] ": " a[2] " " b[2] (a[2]==b[2] ? "" : "   <-- differs")}' | tee $O/form_hash.txt | grep -c differs
for i in 1 2 3; do
  echo "element walk:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "16-byte reads:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu (product) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
