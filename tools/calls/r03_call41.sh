#!/bin/bash
# round 3, call 41: spilling generation streams with wider coalesced passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 ${STEPS:---steps 20} $EXTRA > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], (d.get('parity') or {}).get('timed_steps_identical'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
P
}
STEPS="--steps 24"
run base A=1
EXTRA="--coalesce 3" run spill64_c3 SOPRO_AR_SPILL=1
EXTRA="--coalesce 4" run spill64_c4 SOPRO_AR_SPILL=1
EXTRA="--coalesce 3 --ar-cus 48" run spill48_c3 SOPRO_AR_SPILL=1
EXTRA="--coalesce 3" run base_c3 A=1
EXTRA="--ar-parts 1 --lanes 3" run spill64_p1 SOPRO_AR_SPILL=1
EXTRA="--lanes 3" run spill64_l3 SOPRO_AR_SPILL=1
EXTRA="--ar-cus 40" run spill40 SOPRO_AR_SPILL=1
