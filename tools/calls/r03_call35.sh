#!/bin/bash
# round 3, call 35: where a batch's conditioning runs (SOPRO_PREP_PLACE = bulk | ar | locked), A/B on one box + lane timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03y; O=gpurun_out/r03y
run() { # name, env..., -- bench args
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 $EXTRA > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['roofline']['avg_launch_us'])
P
}
for i in 1 2; do
  run bulk$i SOPRO_PREP_PLACE=bulk
  run ar$i SOPRO_PREP_PLACE=ar
  run locked$i SOPRO_PREP_PLACE=locked
done
EXTRA="--lanes 5" run locked_l5 SOPRO_PREP_PLACE=locked
EXTRA="--lanes 5" run ar_l5 SOPRO_PREP_PLACE=ar
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 > $O/trace.json 2> $O/trace.err
grep -E "step |idle|slot" $O/trace.err | head -60
