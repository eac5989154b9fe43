#!/bin/bash
# round 3, call 22: 32 x 400 frames with and without coalescing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03n; O=gpurun_out/r03n
Q="--frames 400 --steps 12 --warmup 4 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for c in 2 1; do timeout 600 python bench.py $Q --coalesce $c > $O/f400_c$c.json 2> $O/f400_c$c.err; python - "$O/f400_c$c.json" "c$c" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
tail -2 $O/f400_c$c.err
done
