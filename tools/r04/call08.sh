#!/bin/bash
# r04 call 8: where does the full form of the bench die (call 7: SIGSEGV in the bf16 leg)?  faulthandler's stacks.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c08; mkdir -p $O; cd $R
timeout 300 python -X faulthandler bench.py --steps 6 --warmup 4 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 > $O/legs.json 2> $O/legs.err; echo "rc $?"
grep -v "^\[bench" $O/legs.err | head -80
