#!/bin/bash
# round 3, call 39: which CUs a masked stream really uses (census), and whether the throughput partition is bound by its CU count
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
( cd tools/micro && ./mask_census ) | tee $O/mask_census.txt
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
run base A=1
run bulk160 SOPRO_BULK_CUS=160
run bulk128 SOPRO_BULK_CUS=128
run bulk96 SOPRO_BULK_CUS=96
EXTRA="--ar-cus 96" run ar96_bulk128 SOPRO_BULK_CUS=128
EXTRA="--ar-cus 128" run ar128_bulk128 SOPRO_BULK_CUS=128
