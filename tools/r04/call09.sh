#!/bin/bash
# r04 call 9: the exact crashing command of call 7 under faulthandler.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c09; mkdir -p $O; cd $R
timeout 400 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/full.json 2> $O/full.err; echo "rc $?"
grep -v "^\[bench" $O/full.err | head -120
tail -3 $O/full.err
