#!/bin/bash
# round 3, call 19: full GPU suite + driver-form bench on the tree with the MFMA cross-attention, the library profiler, wide AR tiles
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03k; O=gpurun_out/r03k
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03k/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('ok'), d['parity'].get('timed_steps_identical'), d.get('ttfa_ms_p50'))
for e in [d['roofline']]+d['roofline_more']: print(e['kernel'][:40], e.get('avg_launch_us'), e.get('ms_per_step'), e.get('frac'))
for k,v in d['legs'].items(): print(k, v.get('value'), v.get('ms_per_step'))
P
