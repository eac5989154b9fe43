#!/bin/bash
# r05 call 5: steady-state sweep (64 timed steps): jobs per pass 2 / 4 / 6 / 8 x generation partition 48 / 56 / 64 / 80 CUs x lanes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c05; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 64 --warmup 5"
run() {  # name, args
  n=$1; shift
  timeout 300 python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-16s' % '$n', d['value'], d['ms_per_step'], 'warm', d['warmup_run'], d['phase_ms_per_step'], 'ident', d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
run c2
run c4 --coalesce 4
run c4_cus48 --coalesce 4 --ar-cus 48
run c4_cus56 --coalesce 4 --ar-cus 56
run c4_cus80 --coalesce 4 --ar-cus 80
run c6 --coalesce 6
run c8 --coalesce 8
run c8_cus48 --coalesce 8 --ar-cus 48
run c4_l6 --coalesce 4 --lanes 6
run c8_l3 --coalesce 8 --lanes 3
run c2_cus80 --ar-cus 80
run c2_b
uptime
