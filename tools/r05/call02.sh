#!/bin/bash
# r05 call 2: the two tests that failed in call 1 (+ the range tests), then a SCHEDULING sweep on one box: two refinement / decode
# phases at a time on the throughput partition (--bulk-slots 2: never measured), 128-row passes (--coalesce 4), 6 lanes, CU splits.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c02; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_range.py tests/test_gpu_ops.py -m gpu -q -k "range or host_mirror or overflow or badly" --timeout 240 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | cut -c1-240 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
run() {  # name, args
  n=$1; shift
  timeout 300 python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-26s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ident', d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
run base_a --steps 24 --warmup 8
run slots2 --steps 24 --warmup 8 --bulk-slots 2
run lanes6_slots2 --steps 24 --warmup 12 --lanes 6 --bulk-slots 2
run coal4 --steps 40 --warmup 32 --coalesce 4
run coal4_parts1 --steps 40 --warmup 32 --coalesce 4 --ar-parts 1
run coal4_slots2 --steps 40 --warmup 32 --coalesce 4 --bulk-slots 2
run cus80_slots2 --steps 24 --warmup 8 --ar-cus 80 --bulk-slots 2
run base_b --steps 24 --warmup 8
nvidia-smi >/dev/null 2>&1; uptime
