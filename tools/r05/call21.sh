#!/bin/bash
# r05 call 21: the refinement's contractions were tuned at 12800-row passes (two jobs); four jobs per pass = 25600 rows.  Tile A/B of the
# f16 three-pass family in the pipeline (developer library): 128x128 (product), 64x128, 64x64; arg-max form on 64-row tiles.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c21; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 40"
run() {  # name, env..., --, args
  n=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 300 env SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so "${envs[@]}" python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
run base X=1 --
run t64x128 SOPRO_F16X3_TILE=4 --
run t64x64 SOPRO_F16X3_TILE=5 --
run argmax64 SOPRO_ARGMAX_TM=1 --
run base_b X=1 --
run t64x128_b SOPRO_F16X3_TILE=4 --
uptime
