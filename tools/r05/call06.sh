#!/bin/bash
# r05 call 6: pass sizes from the queue depth (--coalesce auto: single-batch first pass, 2 / 4 per pass), deterministic warm-up
# (PipelinedSynthesizer.prepare), conditioning phases in job order - the driver's form and the steady state, against fixed sizes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c06; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
run() {  # name, args
  n=$1; shift
  timeout 300 python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], 'sizes', d['config'].get('pass_sizes'), d['phase_ms_per_step'], 'ident', d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
run d_auto --steps 20 --warmup 5
run d_c2 --steps 20 --warmup 5 --coalesce 2
run d_auto_b --steps 20 --warmup 5
run d_c2_b --steps 20 --warmup 5 --coalesce 2
run s32_auto --steps 32 --warmup 5
run s32_c2 --steps 32 --warmup 5 --coalesce 2
run s64_auto --steps 64 --warmup 5
run s64_c4 --steps 64 --warmup 5 --coalesce 4
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py $Q --steps 20 --warmup 5 > $O/trace.json 2> $O/trace.err
grep -E "  step|idle" $O/trace.err | cut -c1-170
uptime
