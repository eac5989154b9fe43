Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
run() { echo -n "$* -> "; timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'])"; }
run
run --batch 64
run --batch 64 --lanes 3
run --batch 64 --lanes 6
run --batch 48
run --ar-cus 80
run --lanes 5
run --lanes 6 --ar-parts 3 --ar-cus 96
SOPRO_AR_TILES=1x2 run
SOPRO_AR_GRAPH_FRAMES=1 run
SOPRO_AR_GRAPH_FRAMES=16 run
run --voices 1
