#!/bin/bash
# round 3, call 32: tile shapes of the refinement's contractions at 64-row passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03w; O=gpurun_out/r03w
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for t in 0 1 4 0; do SOPRO_F16X3_TILE=$t timeout 300 python bench.py $Q > $O/b$t.json 2> $O/b$t.err; python - $O/b$t.json $t <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('f16x3 tile', sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
except Exception as e: print(sys.argv[2], 'failed', e)
P
done
