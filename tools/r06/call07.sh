#!/bin/bash
# r06 call 7: the long-K equivalence test again; then per-launch tables of one 64 x 200 decode on the 192-CU partition: long-K flow vs the round-5 flow
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c07; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "long_k" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -3 $O/pytest_a.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for v in new:1 old:0; do
  SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so SOPRO_GEMM_8P=${v##*:} DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${v%%:*} -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/${v%%:*}.log 2>&1
  tail -1 $O/${v%%:*}.log
  f=$(find $O/${v%%:*} -name "*kernel_trace.csv" | head -1)
  python $R/tools/r06/decode_table.py $f > $O/${v%%:*}_table.txt; tail -25 $O/${v%%:*}_table.txt
  rm -f $f
done
