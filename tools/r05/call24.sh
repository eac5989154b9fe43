#!/bin/bash
# r05 call 24 ("32-row weight reuse measured"): the AR frame's skinny tile shape at 128-row frames in the pipeline:
# 1x2 (product: 16 rows x two column tiles), 2x1 / 2x2 (32 rows per workgroup: every weight fragment serves two row tiles), 1x1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c24; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 40"
run() {  # name, env..., --, args
  n=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 300 env "${envs[@]}" python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-10s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'frame us', d['roofline']['avg_launch_us'], d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
run t1x2 SOPRO_AR_TILES_WIDE=1x2 --
run t2x1 SOPRO_AR_TILES_WIDE=2x1 --
run t2x2 SOPRO_AR_TILES_WIDE=2x2 --
run t1x1 SOPRO_AR_TILES_WIDE=1x1 --
run t1x2_b SOPRO_AR_TILES_WIDE=1x2 --
run t2x1_b SOPRO_AR_TILES_WIDE=2x1 --
uptime
