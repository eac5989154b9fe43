#!/bin/bash
# r06 call 1: the tree after the record / ADVICE fixes (GPU suite), the driver's form (is the compact line parsed?), and item 3's
# saturation probe (clock / power beside decode at several partition sizes).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c01; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
cp bench_full.json $O/bench_full.json 2>/dev/null
python - <<P
import json
t=open('$O/bench_line.json').read().strip().splitlines()[-1]
d=json.loads(t); print('line bytes', len(t), 'value', d['value'], d['ms_per_step'], d['phase_ms_per_step'])
print('roofline', d['roofline']); print([ (e['kernel'], e['ms_per_step'], e.get('frac_of_pass_ceiling')) for e in d['roofline_more']])
print('cpu', d['cpu_baseline']); print('parity', d['parity'])
P
timeout 900 python tools/r06/saturation_probe.py > $O/saturation.txt 2> $O/saturation.err; echo "probe rc $?"; cat $O/saturation.txt | cut -c1-300
uptime
