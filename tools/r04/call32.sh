#!/bin/bash
# r04 call 32: the 128-channel residual block's tile request with the non-temporal hint (one-off patch of seanet_res.hip: its fp32
# tile loads through bulk_load16(.., true)) against the product: pipeline A/B, fp32.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c32; mkdir -p $O; cd $R
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in prod nt prod nt; do
  L=""; [ $v = nt ] && L=$R/tools/micro/libsopro_nt_resin.so
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q >> $O/f32_$v.json 2>> $O/f32_$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c32'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
