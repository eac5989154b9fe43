#!/bin/bash
# round 3, call 24: conditioning + reference preparation as library sequences
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03p; O=gpurun_out/r03p
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['host_cpu_s_per_step'])
P
