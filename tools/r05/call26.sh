#!/bin/bash
# r05 call 26: the GPU suite three times in a row on the final library (flake screen: the recording invalidation of call 17 was a 1-in-20 event)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c26; mkdir -p $O; cd $R
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu_$i.log 2>&1; echo "run $i rc $?"
  grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu_$i.log | cut -c1-200 | tail -5
done
uptime
