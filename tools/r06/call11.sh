#!/bin/bash
# r06 call 11: 128 x 128 arg-max tiles for the refinement's stage heads: token fixtures, A/B of one refinement pass, whole suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c11; mkdir -p $O; cd $R
for i in 1 2; do
  for v in new:0 old:1; do
    SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so SOPRO_ARGMAX_TN=${v##*:} timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement | sed "s/^/${v%%:*} /"
  done
done
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
