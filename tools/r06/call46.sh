#!/bin/bash
# r06 call 46: every launch of one 128 x 200 refinement pass and of one 64 x 200 decode on the final library (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c46; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
NAR_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/nar -o t -- python $R/tools/r06/nar_run.py 192 4 > $O/nar.log 2>&1
f=$(find $O/nar -name "*kernel_trace.csv" | head -1); python $R/tools/r06/nar_table.py $f > $O/nar_table.txt; tail -16 $O/nar_table.txt; rm -f $f
DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/dec.log 2>&1
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; tail -22 $O/dec_table.txt; rm -f $f
