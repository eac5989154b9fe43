#!/bin/bash
# r05 call 9: pass-size patterns in the driver's 20-step form (which sizes, in which order, fill and drain the pipeline best?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c09; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 20 --warmup 5"
run() {  # name, args
  n=$1; shift
  timeout 300 python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-12s' % '$n', d['value'], d['ms_per_step'], 'sizes', d['config'].get('pass_sizes'), d['phase_ms_per_step'])
except Exception as e: print('$n ERR', e)
P
}
run c2 --coalesce 2
run p5555 --coalesce 5,5,5,5
run p4x5 --coalesce 4
run p3344 --coalesce 3,3,3,3,4,4
run p2244 --coalesce 2,2,4,4,4,4
run p4466 --coalesce 4,4,6,6
run p6644 --coalesce 6,6,4,4
run p2233 --coalesce 2,2,3,3,3,3,2,2
run p3333 --coalesce 3,3,3,3,3,3,2
run p4422 --coalesce 4,4,4,4,2,2
run p2266 --coalesce 2,2,6,6,4
run c2_b --coalesce 2
run p5555_b --coalesce 5,5,5,5
uptime
