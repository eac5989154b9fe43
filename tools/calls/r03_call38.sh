#!/bin/bash
# round 3, call 38: the AR frame's weight stream with non-temporal requests (SOPRO_AR_W_NT=1): what it does to the throughput
# partition's kernels that share the XCDs' L2s, on the range layout with 64 / 96 / 128 CUs and on whole XCDs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
SOPRO_AR_W_NT=1 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -x -q -m gpu -k "skinny or teacher or coalesc or sequential or full200 or 32x200" 2>&1 | tail -2
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 ${STEPS:---steps 20} $EXTRA > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], (d.get('parity') or {}).get('timed_steps_identical'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
P
}
run base A=1
run wnt SOPRO_AR_W_NT=1
run base_b A=1
run wnt_b SOPRO_AR_W_NT=1
EXTRA="--ar-cus 96" run c96 A=1
EXTRA="--ar-cus 96" run c96_wnt SOPRO_AR_W_NT=1
EXTRA="--ar-cus 128" run c128_wnt SOPRO_AR_W_NT=1
run xcd_wnt SOPRO_AR_W_NT=1 SOPRO_PART_LAYOUT=xcd
EXTRA="--ar-cus 128" run xcd4w_wnt SOPRO_AR_W_NT=1 SOPRO_PART_LAYOUT=xcd
EXTRA="--ar-cus 128" run xcd4w SOPRO_PART_LAYOUT=xcd
EXTRA="--precision bf16" run bf16_wnt SOPRO_AR_W_NT=1
EXTRA="--precision bf16" run bf16_base A=1
