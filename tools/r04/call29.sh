#!/bin/bash
# (kept as the record of the call: the variant library came from a one-off patch of gload() in csrc/gemm_bf16s.hip - A rows through
#  __builtin_nontemporal_load - that was not kept; profiles/r04_experiments.md section 7)
# r04 call 29: A rows non-temporal only for the LAST column tile of a row tile (tools/micro/build_nt_a.sh) against the product (outputs
# non-temporal, A rows plain): pipeline A/B, fp32 and bf16.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c29; mkdir -p $O; cd $R
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in prod nta prod nta; do
  L=""; [ $v = nta ] && L=$R/tools/micro/libsopro_nt_a2.so
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q >> $O/f32_$v.json 2>> $O/f32_$v.err
done
for v in prod nta; do
  L=""; [ $v = nta ] && L=$R/tools/micro/libsopro_nt_a2.so
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_$v.json 2>> $O/bf16_$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c29'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
grep -i "error\|Traceback" $O/*.err | head
