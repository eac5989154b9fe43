#!/bin/bash
# r06 call 9: tail split: operator test, whole GPU suite, A/B in the driver's form (SOPRO_TAIL_SLOTS=0 = off), decode table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c09; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "tail_split or long_k" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -3 $O/pytest_a.log | cut -c1-300
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
f=$(find $O/dec -name "*kernel_trace.csv" | head -1); python $R/tools/r06/decode_table.py $f > $O/dec_table.txt; tail -24 $O/dec_table.txt; rm -f $f
