#!/bin/bash
# r05 call 23: the bench lines of the final library (after the scratch sharing), stamped with the committed profiles/r05_* kernel stats / PMC summaries; smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c23; mkdir -p $O; cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log | cut -c1-200
bash tools/collect_evidence.sh r05c23 bench 2>&1 | tail -9 | cut -c1-520
grep -h "peak device memory" $O/*.err | cut -c1-160
