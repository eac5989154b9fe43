#!/bin/bash
# r04 call 33: conditioning (text encoder, reference cross-attention, folding of the text operands) on the GENERATION partition's CUs
# instead of the throughput partition's (SOPRO_PREP_ON_AR=1): the throughput half is the longer one now.  Pipeline A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c33; mkdir -p $O; cd $R
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in 0 1 0 1; do
  SOPRO_PREP_ON_AR=$v timeout 300 python bench.py $Q >> $O/f32_prep$v.json 2>> $O/f32_prep$v.err
done
for v in 0 1; do
  SOPRO_PREP_ON_AR=$v timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_prep$v.json 2>> $O/bf16_prep$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c33'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
grep -i "error\|Traceback" $O/*.err | head
