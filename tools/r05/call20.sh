#!/bin/bash
# r05 call 20: device memory: the pipeline's lanes decode in ONE set of scratch buffers (one throughput slot = one decode at a
# time) and the fused last level no longer gets a buffer for the activation it keeps on the CU.  Whole suite, then the census and the A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c20; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for n in shared:1 own:0 shared_b:1 own_b:0; do
  SOPRO_SHARE_SCRATCH=${n##*:} SOPRO_BENCH_MEMCENSUS=1 timeout 300 python bench.py $Q > $O/${n%%:*}.json 2> $O/${n%%:*}.err
  python - <<P
import json
d=json.loads(open('$O/${n%%:*}.json').read().strip().splitlines()[-1])
print('%-10s' % '${n%%:*}', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
  grep -E "codec.ws|allocated" $O/${n%%:*}.err | cut -c1-200
done
SOPRO_BENCH_MEMCENSUS=1 timeout 300 python bench.py $Q --precision bf16 > $O/bf16.json 2> $O/bf16.err; grep -E "allocated" $O/bf16.err | cut -c1-200
uptime
