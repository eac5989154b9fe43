#!/bin/bash
# r05 call 11: driver's form, a smaller FIRST pass in front of 4-job passes (the throughput partition's first work arrives sooner)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c11; mkdir -p $O; cd $R
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
run c4_a --coalesce 4
run p244442 --coalesce 2,4,4,4,4,2
run p24446 --coalesce 2,4,4,4,6
run p34445 --coalesce 3,4,4,4,5
run p26444 --coalesce 2,6,4,4,4
run p24455 --coalesce 2,4,4,5,5
run p144443 --coalesce 1,4,4,4,4,3
run c4_b --coalesce 4
run p244442_b --coalesce 2,4,4,4,4,2
run p34445_b --coalesce 3,4,4,4,5
uptime
