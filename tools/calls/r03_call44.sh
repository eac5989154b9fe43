#!/bin/bash
# round 3, call 44: split window attention with the exp2-domain / lazy-reference softmax and uniform value-row addresses
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03g; O=$GRAFT_REPO_ROOT/gpurun_out/r03g
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py tests/test_gpu_stages.py -x -q -m gpu -k "attention or full200 or full400 or stream or 32x200 or profiler" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4 -o l4 -- $B --steps 8 --warmup 5 > $O/l4.log 2>&1
grep -E "attn_mfma|Name" $O/l4/l4_kernel_stats.csv | cut -c1-200
rm -f $O/l4/*_kernel_trace.csv
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 $B --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))"; done
