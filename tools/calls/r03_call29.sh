#!/bin/bash
# round 3, call 29: smaller GEMM tiles (fewer registers, more waves per SIMD) in the pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03u; O=gpurun_out/r03u
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for t in 0 4 5 2; do SOPRO_GEMM_TILE=$t timeout 300 python bench.py $Q > $O/b$t.json 2> $O/b$t.err; python - $O/b$t.json $t <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('tile', sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
except Exception as e: print(sys.argv[2], 'failed', e)
P
done
