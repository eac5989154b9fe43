#!/bin/bash
# r05 call 13: RoPE in the qkv contraction's epilogue (SOPRO_EPI_ROPE): operator test, the Mimi fixtures, then an A/B of the pipeline
# with the developer library (libsopro_hip_dev.so reads SOPRO_ROPE_FUSE)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c13; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -k "rope or mimi or decode or stream or full or e2e or stages or uptail" --timeout 240 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5"
run() {  # name, env, args
  n=$1; e=$2; shift; shift
  timeout 300 env SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so $e python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-12s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
except Exception as e: print('$n ERR', e)
P
}
for i in 1 2 3; do
run sep_$i SOPRO_ROPE_FUSE=0 --steps 40
run fused_$i SOPRO_ROPE_FUSE=1 --steps 40
done
uptime
