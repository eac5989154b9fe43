#!/bin/bash
# round 3, call 16: tail / res128 with the next tile's request in flight; attention probe; decoder probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03h; O=gpurun_out/r03h
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" ; tail -3 $O/pytest_a.log
timeout 200 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_probe.txt
timeout 300 python tools/mimi_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mimi_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'])
for e in [d['roofline']]+d['roofline_more']: print(e['kernel'][:40], e.get('avg_launch_us'), e.get('ms_per_step'), e.get('frac'))
P
