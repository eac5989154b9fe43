#!/bin/bash
# r06 call 52: arg-max partials reduced by eight lanes per head (product library) against one thread per head (developer library of call 51):
# refinement pass x 3, arg-max / refinement tests, launch table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c52; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
for i in 1 2 3; do
  echo "one thread per head:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "eight lanes:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu (product) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
cd /tmp && export TMPDIR=/tmp
NAR_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/nar -o t -- python $R/tools/r06/nar_run.py 192 4 > $O/nar.log 2>&1
f=$(find $O/nar -name "*kernel_trace.csv" | head -1); python $R/tools/r06/nar_table.py $f > $O/nar_table.txt; tail -14 $O/nar_table.txt | head -13; rm -f $f
