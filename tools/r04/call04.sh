#!/bin/bash
# r04 call 4: GPU suite on the unfolded-key frame (default on), the bf16 mode's quality numbers, and the A/B of the frame
# (SOPRO_AR_KUNFOLD=0 = folded keys, the round-3 frame) in both modes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c04; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 200 python -m pytest tests/test_gpu_bf16_mode.py -q -s -k "quality or mimi_decode" --timeout 150 2>&1 | grep -i "bf16 mode" > $O/bf16_quality.txt; cat $O/bf16_quality.txt
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for uk in 0 1 0 1; do
  SOPRO_AR_KUNFOLD=$uk timeout 300 python bench.py $Q >> $O/f32_uk$uk.json 2>> $O/f32_uk$uk.err
done
for uk in 0 1; do
  SOPRO_AR_KUNFOLD=$uk timeout 300 python bench.py $Q --precision bf16 > $O/bf16_uk$uk.json 2> $O/bf16_uk$uk.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c04'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'cpu/step', d['host_cpu_s_per_step'], d['parity'].get('timed_steps_identical'))
        except Exception as e: print(f, 'ERR', e)
P
