#!/bin/bash
# r06 call 19: whole GPU suite + driver-form bench (x2) on the library with the res128 prefetch / addressing changes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c19; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2 3; do
  timeout 300 python bench.py $Q > $O/b_$i.json 2> $O/b_$i.err
  python - <<P
import json
d=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
done
timeout 300 python bench.py $Q --precision bf16 > $O/bf16.json 2> $O/bf16.err
python - <<P
import json
d=json.loads(open('$O/bf16.json').read().strip().splitlines()[-1])
print('bf16', d['value'], d['ms_per_step'], d['phase_ms_per_step'])
P
