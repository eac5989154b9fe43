#!/bin/bash
# r04 call 1: GPU suite on the round's first changes (oracle-based serving tests, strict fixtures, sampler / speaker fixtures,
# odd-K barrier, bf16 K'/V' + rings) and two A/B pairs of the bench: host wait mode, bf16 state in memory.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c01; mkdir -p $O; cd $R
nproc > $O/host.txt; uptime >> $O/host.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for w in 0 1; do
  SOPRO_BLOCKING_WAIT=$w timeout 300 python bench.py $Q > $O/f32_wait$w.json 2> $O/f32_wait$w.err
done
for st in 0 1; do
  SOPRO_BF16_STATE=$st timeout 300 python bench.py $Q --precision bf16 > $O/bf16_state$st.json 2> $O/bf16_state$st.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c01'
for f in sorted(glob.glob(O+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], 'ar us', d['roofline']['avg_launch_us'] if d.get('roofline') else None, d['parity'].get('timed_steps_identical'))
    except Exception as e: print(f, 'ERR', e)
P
