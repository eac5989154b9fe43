#!/bin/bash
# r05 call 1: the GPU suite on the round's first changes (stages hand over in place - no runtime copies between the library's launch
# sequences -, f16x3 range guard + badly scaled reference fixtures, C-side chunked decode, hardened checkpoint loader), smoke, then
# the bench in the driver's form at two CU splits and a kernel trace of the pipelined run (are any at::native / copyBuffer rows left?).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c01; mkdir -p $O; cd $R
nproc > $O/host.txt; uptime >> $O/host.txt
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -16
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log | cut -c1-200
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for cus in 64 80 64; do
  timeout 300 python bench.py $Q --ar-cus $cus > $O/f32_cus$cus.json 2> $O/f32_cus$cus.err || tail -3 $O/f32_cus$cus.err | cut -c1-300
  python - <<P
import json
try:
    d=json.loads(open('$O/f32_cus$cus.json').read().strip().splitlines()[-1])
    print('cus $cus', d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ar us', (d.get('roofline') or {}).get('avg_launch_us'), 'ident', d['parity'].get('timed_steps_identical'), 'fallbacks', d['parity'].get('f16_range_fallbacks'))
except Exception as e: print('cus $cus ERR', e)
P
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4 -o l4 -- python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 8 --warmup 5 > $O/l4.log 2>&1
rm -f $O/l4/*_kernel_trace.csv
cd $R
python - <<P
import csv
rows=list(csv.DictReader(open('$O/l4/l4_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
bad=[r for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'elementwise_kernel' in r['Name']]
print('kernels', len(rows), 'total ms', round(tot/1e6,1), 'torch/runtime rows', len(bad), 'ms', round(sum(float(r['TotalDurationNs']) for r in bad)/1e6,2), 'calls', sum(int(r['Calls']) for r in bad))
for r in sorted(bad, key=lambda r:-float(r['TotalDurationNs']))[:8]: print('  ', r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs']))[:14]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e3,1))
P
