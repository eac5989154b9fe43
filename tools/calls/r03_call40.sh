#!/bin/bash
# round 3, call 40: generation streams that may spill onto the throughput partition's CUs (the reverse is still masked out)
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
run base A=1
run spill64 SOPRO_AR_SPILL=1
EXTRA="--ar-cus 48" run spill48 SOPRO_AR_SPILL=1
EXTRA="--ar-cus 32" run spill32 SOPRO_AR_SPILL=1
EXTRA="--ar-cus 16" run spill16 SOPRO_AR_SPILL=1
EXTRA="--ar-cus 32 --ar-parts 3 --lanes 6" run spill32_p3 SOPRO_AR_SPILL=1
