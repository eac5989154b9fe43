#!/bin/bash
# round 3, call 37: generation partition on fewer XCDs - what the AR phase needs there (tile shapes, concurrent phases, hybrid masks)
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
run range A=1
run xcd SOPRO_PART_LAYOUT=xcd
run xcd4 SOPRO_PART_LAYOUT=xcd4
run xcd_2x2 SOPRO_PART_LAYOUT=xcd SOPRO_AR_TILES_WIDE=2x2
run xcd_2x1 SOPRO_PART_LAYOUT=xcd SOPRO_AR_TILES_WIDE=2x1
run xcd_1x1 SOPRO_PART_LAYOUT=xcd SOPRO_AR_TILES_WIDE=1x1
EXTRA="--ar-parts 3 --lanes 6" run xcd_p3_l6 SOPRO_PART_LAYOUT=xcd
EXTRA="--ar-parts 3 --lanes 5" run xcd_p3_l5 SOPRO_PART_LAYOUT=xcd
EXTRA="--ar-cus 96" run xcd3_96 SOPRO_PART_LAYOUT=xcd
EXTRA="--ar-cus 96" run xcd4_96 SOPRO_PART_LAYOUT=xcd4
EXTRA="--lanes 5" run xcd_l5 SOPRO_PART_LAYOUT=xcd
EXTRA="--coalesce 1 --ar-parts 3 --lanes 6" run xcd_c1_p3 SOPRO_PART_LAYOUT=xcd
STEPS="--steps 40" run range_s40 A=1
STEPS="--steps 40" run xcd_s40 SOPRO_PART_LAYOUT=xcd
EXTRA="--precision bf16" run bf16_xcd SOPRO_PART_LAYOUT=xcd
EXTRA="--precision bf16" run bf16_range A=1
