#!/bin/bash
# r05 call 14: comb form of the depthwise convolution: equivalence test + the token-path fixtures, then A/B in the pipeline (dev library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c14; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -k "dwconv or nar or refine or full or prepare_conditioning or range or stages or pipeline" --timeout 240 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5"
run() {  # name, env, args
  n=$1; e=$2; shift; shift
  timeout 300 env SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so $e python bench.py $Q "$@" > $O/$n.json 2> $O/$n.err || tail -4 $O/$n.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
    print('%-12s' % '$n', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d['parity'].get('rank_output_sha16'))
except Exception as e: print('$n ERR', e)
P
}
for i in 1 2 3; do
run old_$i SOPRO_DWCONV_COMB=0 --steps 40
run comb_$i SOPRO_DWCONV_COMB=1 --steps 40
done
uptime
