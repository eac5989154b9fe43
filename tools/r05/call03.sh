#!/bin/bash
# r05 call 3: the driver's form (--steps 20 --warmup 5) under pass sizes of 2 / 3 / 4 / 5 coalesced jobs, lane counts and the
# warm-up rule (round 4: one pass per lane - every lane's recording pass fell into the timed region; now two)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c03; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 20 --warmup 5"
run() {  # name, env..., -- args
  n=$1; shift
  timeout 300 env "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-22s' % '$n', d['value'], d['ms_per_step'], 'warm', d['warmup_run'], d['phase_ms_per_step'], 'ident', d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
B="python bench.py $Q"
run c2_warm1 SOPRO_BENCH_WARM_PASSES=1 $B
run c2 A=1 $B
run c3 A=1 $B --coalesce 3
run c4 A=1 $B --coalesce 4
run c5 A=1 $B --coalesce 5
run c4_l5 A=1 $B --coalesce 4 --lanes 5
run c4_l6 A=1 $B --coalesce 4 --lanes 6
run c3_l6 A=1 $B --coalesce 3 --lanes 6
run c4_cus80 A=1 $B --coalesce 4 --ar-cus 80
run c4_wide11 SOPRO_AR_TILES_WIDE=1x1 $B --coalesce 4
run c4_wide22 SOPRO_AR_TILES_WIDE=2x2 $B --coalesce 4
run c4_wide21 SOPRO_AR_TILES_WIDE=2x1 $B --coalesce 4
run c4_wide14 SOPRO_AR_TILES_WIDE=1x4 $B --coalesce 4
run c2_b A=1 $B
run c4_b A=1 $B --coalesce 4
run c4_40 A=1 python bench.py --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 40 --warmup 5 --coalesce 4
uptime
