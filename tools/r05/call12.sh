#!/bin/bash
# r05 call 12: the per-pass preparation (conditioning, text folding) on the generation partition's CUs now that it has slack
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c12; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5"
run() {  # name, env, args
  n=$1; e=$2; shift; shift
  timeout 300 env $e python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], 'sizes', d['config'].get('pass_sizes')[:3], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
for i in 1 2; do
run d_bulk_$i SOPRO_PREP_ON_AR=0 --steps 20
run d_ar_$i SOPRO_PREP_ON_AR=1 --steps 20
done
run s_bulk SOPRO_PREP_ON_AR=0 --steps 64
run s_ar SOPRO_PREP_ON_AR=1 --steps 64
run s_ar_c6 SOPRO_PREP_ON_AR=1 --steps 64 --coalesce 6
run s_bulk_c6 SOPRO_PREP_ON_AR=0 --steps 64 --coalesce 6
uptime
