#!/bin/bash
# r06 call 6: the long-K form in the product library: operator equivalence, the decoder's parity fixtures, then the A/B in the
# driver's form (dev library: SOPRO_GEMM_8P=0 = the round-5 flow) alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_uptail.py tests/test_gpu_range.py tests/test_gpu_hostblocks.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "long_k or mimi or split_plane or range or hostblock or recorded or decode_parts or scheduler" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -5 $O/pytest_a.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  for v in new:1 old:0; do
    SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so SOPRO_GEMM_8P=${v##*:} timeout 300 python bench.py $Q > $O/${v%%:*}_$i.json 2> $O/${v%%:*}_$i.err
    python - <<P
import json
d=json.loads(open('$O/${v%%:*}_$i.json').read().strip().splitlines()[-1])
print('%-6s' % '${v%%:*}', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
  done
done
uptime
