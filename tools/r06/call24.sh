#!/bin/bash
# r06 call 24: window attention with the key / value tiles shared by a workgroup's waves: operator tests, decode checksum (bit-identical?), A/B of the decode
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c24; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention or mimi or decode or e2e or stream or full" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -4 $O/pytest_a.log | cut -c1-300
for i in 1 2; do
  for v in shared:1 perwave:0; do
    SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so SOPRO_ATTN_SHARED=${v##*:} timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64" | sed "s/^/${v%%:*} /"
  done
done
cd /tmp && export TMPDIR=/tmp
DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/dec.log 2>&1
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; grep -i "attn" $O/dec_table.txt | tail -3; rm -f $f
