timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -x 2>&1 | grep -E "^E |passed|failed|FAILED" | head -12
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --warmup 5"
run() { echo -n "$* -> "; timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity']['timed_steps_identical'])"; }
run --steps 20 --coalesce 1
run --steps 20 --coalesce 2
run --steps 20 --coalesce 3
run --steps 20 --coalesce 4
run --steps 32 --coalesce 2
run --steps 32 --coalesce 4
run --steps 20 --coalesce 2 --lanes 3
run --steps 20 --coalesce 2 --lanes 6
