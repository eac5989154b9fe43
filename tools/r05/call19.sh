#!/bin/bash
# r05 call 19: now that the run is bound by the throughput partition alone (the generation partition has ~35 % slack at four jobs per
# pass), do the round-4 levers that traded generation time for throughput time pay?  (a) 128x128 tiles on eight waves (SOPRO_GEMM_W8=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c19; mkdir -p $O; cd $R
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
run base X=1 --
run w8 SOPRO_GEMM_W8=1 --
run base_b X=1 --
run w8_b SOPRO_GEMM_W8=1 --
run w8_cus56 SOPRO_GEMM_W8=1 -- --ar-cus 56
run base_bf16 X=1 -- --precision bf16
run w8_bf16 SOPRO_GEMM_W8=1 -- --precision bf16
uptime
