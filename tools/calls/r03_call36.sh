#!/bin/bash
# round 3, call 36: split-bf16 window attention of the decoder (tests + A/B), non-temporal K'/V' reads of the AR cross-attention,
# generation partition as whole XCDs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03y; O=gpurun_out/r03y
S=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "tests: $(( $(date +%s) - S )) s"; S=$(date +%s)
SOPRO_XATTN_NT=1 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -x -q -m gpu -k "xattn or teacher or coalesc or sequential" 2>&1 | tail -2
echo "nt tests: $(( $(date +%s) - S )) s"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 $EXTRA > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('timed_steps_identical'))
P
}
for i in 1 2; do
  run exact$i SOPRO_ATTN_SPLIT=0
  run split$i SOPRO_ATTN_SPLIT=1
  run split_nt$i SOPRO_XATTN_NT=1
  run split_xcd$i SOPRO_PART_LAYOUT=xcd
done
run split_nt_xcd SOPRO_XATTN_NT=1 SOPRO_PART_LAYOUT=xcd
EXTRA="--precision bf16"
run bf16_exact SOPRO_ATTN_SPLIT=0
run bf16_split3 SOPRO_ATTN_SPLIT=1
run bf16_split1 SOPRO_ATTN_PASSES=1
run bf16_split1_nt SOPRO_ATTN_PASSES=1 SOPRO_XATTN_NT=1
