#!/bin/bash
# r05 call 10: driver's form, 2 against 4 jobs per pass, alternating (is the +3 % of call 9 real?), and 5 lanes for the five passes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c10; mkdir -p $O; cd $R
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
for i in 1 2 3; do
run c2_$i --coalesce 2
run c4_$i --coalesce 4
done
run c4_l5 --coalesce 4 --lanes 5
run c4_l3 --coalesce 4 --lanes 3
run c32_def --steps 32 --warmup 2
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py $Q --coalesce 4 > $O/trace.json 2> $O/trace.err
grep -E "  step|idle" $O/trace.err | cut -c1-170
uptime
