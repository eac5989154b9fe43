#!/bin/bash
# round 3, call 33: refinement on 128x128 tiles for many-row passes: token parity tests + bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03x; O=gpurun_out/r03x
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_pipeline.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 > $O/b$i.json 2>/dev/null; python - $O/b$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
done
