O=gpurun_out/r03b; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest.log 2>&1
timeout 120 python tools/sampler_timeline.py > $O/sampler_timeline.txt 2>&1
SOPRO_AR_GRAPH_FRAMES=1 timeout 120 python tools/ar_probe.py 32 200 > $O/ar_probe_g1.txt 2>&1
SOPRO_AR_GRAPH_FRAMES=8 timeout 120 python tools/ar_probe.py 32 200 > $O/ar_probe_g8.txt 2>&1
SOPRO_AR_GRAPH_FRAMES=25 timeout 120 python tools/ar_probe.py 32 200 > $O/ar_probe_g25.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -12 $O/pytest.log; cat $O/sampler_timeline.txt; tail -3 $O/ar_probe_g1.txt $O/ar_probe_g8.txt $O/ar_probe_g25.txt; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ttfa', d['ttfa_ms_p50'], '| roofline', r['achieved'], r['frac'], 'us', r['avg_launch_us'], r.get('isolated_whole_chip'), '| parity', d['parity'].get('ok'), d['parity']['timed_steps_identical'], '| cpu', (d.get('cpu_baseline') or {}))"
tail -5 $O/bench.err
