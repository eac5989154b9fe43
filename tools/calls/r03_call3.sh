O=gpurun_out/r03c; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -6 $O/pytest.log; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ttfa', d['ttfa_ms_p50'], d['cpu_ttfa_ms_p50'], 'hostcpu', d['host_cpu_s_per_step'])
print('roofline', r['kernel'][:30], r['achieved'], r['frac'], 'us', r['avg_launch_us'], r.get('isolated_whole_chip'))
print('parity', d['parity'].get('ok'), d['parity']['timed_steps_identical'])
print(json.dumps(d['legs'], indent=1))
print(d['kernel_families']); print(d['roofline_dropped'])
for e in d['roofline_more']: print(e['kernel'][:40], e['achieved'], e['frac'], e['avg_launch_us'], e.get('ms_per_step'), e.get('gpu_bound_samples'), e.get('samples'))
"
tail -25 $O/bench.err
