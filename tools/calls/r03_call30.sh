#!/bin/bash
# round 3, call 30: two lanes in the throughput phase at once / more lanes, with the coalesced passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03v; O=gpurun_out/r03v
Q="--steps 24 --warmup 6 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for cfg in "4 1" "4 2" "6 2" "6 1" "5 2" "8 2"; do set -- $cfg
  timeout 300 python bench.py $Q --lanes $1 --bulk-slots $2 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - $O/b_$1_$2.json "lanes $1 bulk-slots $2" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'])
except Exception as e: print(sys.argv[2], 'failed', e)
P
done | tee $O/sweep.txt
