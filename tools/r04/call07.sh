#!/bin/bash
# r04 call 7: the driver's full form once (legs, CPU baseline, TTFA under the blocking wait mode), and the A/B of the arg-max
# head contraction's tile height.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c07; mkdir -p $O; cd $R
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err ) 2> $O/driver_form.time; cat $O/driver_form.time
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for tm in 1 0 1 0; do
  SOPRO_ARGMAX_TM=$tm timeout 300 python bench.py $Q >> $O/f32_argmax_tm$tm.json 2>> $O/f32_argmax_tm$tm.err
done
SOPRO_BLOCKING_WAIT=0 timeout 300 python bench.py --steps 4 --warmup 4 --no-legs --no-cpu-baseline --profile-steps 0 --ttfa-runs 50 > $O/ttfa_spin.json 2> $O/ttfa_spin.err
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c07'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], 'ttfa', d.get('ttfa_ms_p50'), d.get('host_wait'), d['parity'].get('timed_steps_identical'), d['parity'].get('ok'))
            if d.get('legs'): print('   legs', {k:(v.get('value'), v.get('quality',{}).get('waveform_snr_db')) for k,v in d['legs'].items() if isinstance(v,dict)})
            if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('reference_ratio',{}).get('estimated_reference_value_here'))
            r=d.get('roofline');
            if r: print('   roofline', r['kernel'][:30], r['achieved'], r['frac'], r['avg_launch_us'], r.get('traffic_ratio'), r.get('traffic_rows_per_launch'))
        except Exception as e: print(f, 'ERR', e)
P
