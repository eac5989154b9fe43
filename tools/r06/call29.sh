#!/bin/bash
# r06 call 29: seanet_uptail without its two scratch reloads per tile (per-lane constants read from LDS; the second residual convolution's
# second-substep fragments requested under the first substep): kernel tests, decode time + checksum, decode table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c29; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_uptail.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py tests/test_gpu_bf16_mode.py -m gpu -q --timeout 300 -p no:cacheprovider -k "seanet or mimi or decode or uptail or e2e or full or bf16" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -4 $O/pytest_a.log | cut -c1-300
for i in 1 2 3; do timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"; done
cd /tmp && export TMPDIR=/tmp
DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/dec.log 2>&1
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; tail -20 $O/dec_table.txt | head -8; rm -f $f
