#!/bin/bash
# round 3, call 34 (re-entry after the container was re-created): the whole GPU suite + smoke + the driver's bench form on HEAD
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03y; O=gpurun_out/r03y
S=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "tests: $(( $(date +%s) - S )) s"; S=$(date +%s)
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
echo "smoke: $(( $(date +%s) - S )) s"; S=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
echo "bench rc $? : $(( $(date +%s) - S )) s"
python - $O/bench_driver_form.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('ok'), d['parity'].get('timed_steps_identical'), d['roofline']['frac'])
print({k:v.get('value') for k,v in d['legs'].items()})
P
