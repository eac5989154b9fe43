#!/bin/bash
# r05 call 25: a shorter ramp for the 20-step driver form?  SOPRO_RAMP_SLOTS=1: every lane's first generation phase starts at once on a quarter
# of the chip (no partition lock) while nothing has reached the throughput partition; with equal passes and with smaller first passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c25; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
run() {  # name, env..., --, args
  n=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 300 env "${envs[@]}" python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['config'].get('pass_sizes'), d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
run base X=1 --
run ramp4 SOPRO_RAMP_SLOTS=1 --
run ramp_2244 SOPRO_RAMP_SLOTS=1 -- --coalesce 2,2,4,4,4,4
run ramp_1124 SOPRO_RAMP_SLOTS=1 -- --coalesce 1,1,2,4,4,4,4
run ramp_3344 SOPRO_RAMP_SLOTS=1 -- --coalesce 3,3,3,3,4,4
run list_2244 X=1 -- --coalesce 2,2,4,4,4,4
run base_b X=1 --
run ramp4_b SOPRO_RAMP_SLOTS=1 --
uptime
