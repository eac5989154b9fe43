#!/bin/bash
# round 3, call 46: a pipeline that runs dry gives its last refinement / decode phases the whole chip (SOPRO_DRAIN_WHOLE)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03h; O=gpurun_out/r03h
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in 0 1 0 1; do SOPRO_DRAIN_WHOLE=$v timeout 200 $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('drain_whole=$v', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))"; done
