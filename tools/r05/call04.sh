#!/bin/bash
# r05 call 4: lane / phase timelines of the driver's form at 2 and 4 coalesced jobs per pass (where does the 20-step form idle?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c04; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 20 --warmup 5"
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py $Q > $O/c2.json 2> $O/c2.err
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py $Q --coalesce 4 > $O/c4.json 2> $O/c4.err
grep -E "step|idle|timed region" $O/c2.err | cut -c1-200
echo ----
grep -E "step|idle|timed region" $O/c4.err | cut -c1-200
