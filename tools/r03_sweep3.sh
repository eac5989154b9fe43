Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --warmup 5 --steps 32"
run() { echo -n "$* -> "; timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'])"; }
run
run --ar-cus 48
run --ar-cus 32
run --ar-cus 40
run --bulk-slots 2
run --lanes 5
run --ar-cus 48 --lanes 5
run --coalesce 3 --lanes 6
run --ar-parts 3 --lanes 6
run --ar-parts 1 --lanes 3
