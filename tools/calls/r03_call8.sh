O=gpurun_out/r03f; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -v "^$" $O/pytest.log | tail -25; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ttfa', d['ttfa_ms_p50'], d['cpu_ttfa_ms_p50'], 'hostcpu', d['host_cpu_s_per_step'])
print('roofline', r['kernel'][:30], r['achieved'], r['frac'], 'us', r['avg_launch_us'], r.get('isolated_whole_chip'))
print('parity', d['parity'].get('ok'), d['parity']['timed_steps_identical'])
for k,v in (d['legs'] or {}).items(): print(k, v if not isinstance(v, dict) else {a:b for a,b in v.items() if a in ('value','ms_per_step','quality','host_cpu_s_per_step')})
"
tail -5 $O/bench.err
