#!/bin/bash
# round 3, call 23: bf16 mode's own line (phases, families)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03o; O=gpurun_out/r03o
timeout 600 python bench.py --steps 20 --warmup 5 --precision bf16 --no-legs --no-cpu-baseline --ttfa-runs 0 > $O/bf16.json 2> $O/bf16.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03o/bf16.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'])
for e in [d['roofline']]+d['roofline_more']: print(e['kernel'][:40], e.get('avg_launch_us'), e.get('ms_per_step'), e.get('frac'))
P
