#!/bin/bash
# r04 call 17: CU split between the generation partition and the throughput partition after the fused last level (the throughput
# half got faster): --ar-cus 48 / 56 / 64 / 72, fp32 and bf16 mode.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c17; mkdir -p $O; cd $R
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for cu in 64 56 48 72 64 56; do
  timeout 300 python bench.py $Q --ar-cus $cu >> $O/f32_cu$cu.json 2>> $O/f32_cu$cu.err
done
for cu in 64 56 48 72; do
  timeout 300 python bench.py $Q --ar-cus $cu --precision bf16 >> $O/bf16_cu$cu.json 2>> $O/bf16_cu$cu.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c17'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
        except Exception as e: print(f, 'ERR', e)
P
grep -i "error\|Traceback" $O/*.err | head
