#!/bin/bash
# r04 call 10: the driver's full form with the legs in child processes; the pipeline / bench-rank tests on the close() change.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c10; mkdir -p $O; cd $R
( time timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err ) 2> $O/driver_form.time; echo "rc $?"; cat $O/driver_form.time
grep "^\[bench" $O/driver_form.err | tail -25; grep -v "^\[bench" $O/driver_form.err | head -30
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bench_ranks.py tests/test_gpu_serving.py -x -q --timeout 200 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
python - <<'P'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c10'
d=json.loads(open(O+'/driver_form.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], 'ttfa', d.get('ttfa_ms_p50'), d.get('host_wait'), d['parity'].get('timed_steps_identical'), d['parity'].get('ok'))
print('legs', {k:(v.get('value'), v.get('error')) for k,v in (d.get('legs') or {}).items()})
print('quality', (d['legs'].get('bf16_32x200') or {}).get('quality'))
print('cpu', d.get('cpu_baseline'))
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','avg_launch_us','traffic','traffic_ratio','traffic_rows_per_launch','rows_per_launch','algorithmic_bytes_per_launch') if k in r})
for m in d['roofline_more']: print('  ', m['kernel'][:50], m['achieved'], m.get('frac_of_pass_ceiling'), m['ms_per_step'])
P
