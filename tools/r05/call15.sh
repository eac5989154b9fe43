#!/bin/bash
# r05 call 15: the AR frame's skinny kernel capped at 128 registers (four resident workgroups per CU): does a 48-CU generation
# partition then behave like today's 64, leaving 208 CUs to the bound half?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c15; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 40"
run() {  # name, lib, args
  n=$1; l=$2; shift; shift
  timeout 300 env SOPRO_HIP_LIB=$R/sopro_amd/$l python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
run base64 libsopro_hip.so
run occ4_64 libsopro_hip_occ4.so
run occ4_48 libsopro_hip_occ4.so --ar-cus 48
run occ4_56 libsopro_hip_occ4.so --ar-cus 56
run occ4_48_c2 libsopro_hip_occ4.so --ar-cus 48 --coalesce 2
run base48 libsopro_hip.so --ar-cus 48
run base64_b libsopro_hip.so
run occ4_64_b libsopro_hip_occ4.so
uptime
