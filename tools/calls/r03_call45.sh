#!/bin/bash
# round 3, call 45: the final build - whole GPU suite, then the bench lines that go to profiles/
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03h; O=gpurun_out/r03h
S=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "tests: $(( $(date +%s) - S )) s"; S=$(date +%s)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
timeout 400 python bench.py > $O/bench_line_default.json 2> $O/bench_line_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision bf16 --no-legs --no-cpu-baseline > $O/bench_line_bf16.json 2> $O/bench_line_bf16.err
for f in bench_line bench_line_default bench_line_bf16; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ttfa', d.get('ttfa_ms_p50'), '| roofline', r['achieved'], r['frac'], 'us', r['avg_launch_us'], '| parity ok', d['parity'].get('ok'), d['parity']['timed_steps_identical'], '| legs', {k: v.get('value') for k, v in (d.get('legs') or {}).items()})"; done
echo "bench: $(( $(date +%s) - S )) s"
