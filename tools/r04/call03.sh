#!/bin/bash
# r04 call 3: whole GPU suite (per-test timeout with thread dumps), then the bench A/B of the decoder's bf16 rows, the fp32 line with
# the wait mode taken at process start, and a kernel-stats profile of the bf16 line.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c03; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for mb in 0 1; do
  SOPRO_MIMI_BF16=$mb timeout 300 python bench.py $Q --precision bf16 > $O/bf16_rows$mb.json 2> $O/bf16_rows$mb.err
done
timeout 300 python bench.py $Q > $O/f32.json 2> $O/f32.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --profile-steps 0 --ttfa-runs 30 --lanes 1 --steps 4 --warmup 2 > $O/f32_lanes1_ttfa.json 2> $O/f32_lanes1_ttfa.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof16 -o p -- python $R/bench.py --steps 8 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --precision bf16 > $O/prof16.log 2>&1
rm -f $O/prof16/*_kernel_trace.csv $O/prof16/*/*_kernel_trace.csv
cd $R
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c03'
for f in sorted(glob.glob(O+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], d.get('host_wait'), 'ttfa', d.get('ttfa_ms_p50'), d['parity'].get('timed_steps_identical'))
    except Exception as e: print(f, 'ERR', e)
P
find $O/prof16 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {} | cut -c1-200'
