#!/bin/bash
# round 3, call 28: full suite + bench with the sixteen-wave tail for long inputs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03t; O=gpurun_out/r03t
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 > $O/bench$i.json 2> $O/bench$i.err; python - $O/bench$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], {e['kernel'][:14]: e.get('avg_launch_us') for e in d['roofline_more'] if 'seanet' in e['kernel']})
P
done
