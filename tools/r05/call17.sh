#!/bin/bash
# r05 call 17: the decoder transformer's LayerNorms fused into the contractions either side (sopro_gemm_split_ext.ln_stats):
# operator test, Mimi / stage / pipeline tests, then the pipeline A/B (developer library: SOPRO_LN_FUSE=0 = the separate norm kernels).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c17; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "layernorm or rope or norm" --timeout 240 -p no:cacheprovider 2>&1 | tail -15 | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 40"
run() {  # name, env..., --, args
  n=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 300 env SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so "${envs[@]}" python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-14s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
run fused X=1 --
run unfused SOPRO_LN_FUSE=0 --
run fused_b X=1 --
run unfused_b SOPRO_LN_FUSE=0 --
run fused_bf16 X=1 -- --precision bf16
run unfused_bf16 SOPRO_LN_FUSE=0 -- --precision bf16
uptime
