#!/bin/bash
# r06 call 10: tail split with the scratch carved for many-row decodes: decoder fixtures, A/B, decode table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c10; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  for v in on:384 off:0; do
    SOPRO_TAIL_SLOTS=${v##*:} timeout 300 python bench.py $Q > $O/${v%%:*}_$i.json 2> $O/${v%%:*}_$i.err
    python - <<P
import json
d=json.loads(open('$O/${v%%:*}_$i.json').read().strip().splitlines()[-1])
print('%-6s' % '${v%%:*}', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
  done
done
cd /tmp && export TMPDIR=/tmp
DECODE_EAGER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -o t -- python $R/tools/r06/decode_run.py 192 4 > $O/dec.log 2>&1
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; tail -28 $O/dec_table.txt; rm -f $f
