#!/bin/bash
# r06 call 8: per-launch table of one 128 x 200 refinement pass on the 192-CU partition
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c08; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
NAR_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/nar -o t -- python $R/tools/r06/nar_run.py 192 3 > $O/nar.log 2>&1
grep refinement $O/nar.log
f=$(find $O/nar -name "*kernel_trace.csv" | head -1)
python $R/tools/r06/nar_table.py $f > $O/nar_table.txt; cat $O/nar_table.txt | cut -c1-180
rm -f $f
