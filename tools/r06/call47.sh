#!/bin/bash
# r06 call 47: window attention with its wave-uniform values read through the first lane (scalar row bases / tile loop): product library
# against the developer library (still the previous attention kernel), attention tests, decode x 3 each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c47; mkdir -p $O; cd $R
D="SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "attention or attn or mimi or decode or stream or e2e or full" > $O/pytest_a.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_a.log | cut -c1-200
for i in 1 2 3; do
  echo "old:"; env $D timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
  echo "new:"; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
done
cd /tmp && export TMPDIR=/tmp
DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/dec.log 2>&1
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; grep "attn_mfma_split\|uptail" $O/dec_table.txt | tail -3; rm -f $f
