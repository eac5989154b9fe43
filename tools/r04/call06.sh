#!/bin/bash
# r04 call 6: GPU suite (C loader engine, ramp, unfolded keys default in fp32 only) + ramp A/B in both modes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c06; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for rp in 0 1 0 1; do
  SOPRO_PIPE_RAMP=$rp timeout 300 python bench.py $Q >> $O/f32_ramp$rp.json 2>> $O/f32_ramp$rp.err
done
for rp in 0 1; do
  SOPRO_PIPE_RAMP=$rp timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_ramp$rp.json 2>> $O/bf16_ramp$rp.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c06'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], d['parity'].get('timed_steps_identical'))
        except Exception as e: print(f, 'ERR', e)
P
