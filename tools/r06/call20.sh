#!/bin/bash
# r06 call 20: fast-erf GELU in the throughput phases' contraction epilogues: per-shape cost, refinement pass time + checksum (tokens), whole suite, bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c20; mkdir -p $O; cd $R
timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/ff_cost.txt | cut -c1-200
for i in 1 2; do timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; done
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  timeout 300 python bench.py $Q > $O/b_$i.json 2> $O/b_$i.err
  python - <<P
import json
d=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
done
