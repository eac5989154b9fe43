#!/bin/bash
# r04 call 41: requests dealt out between the MFMAs for the one-pass contraction and the f16x3 (refinement) family, three-pass bf16x3
# as before (the new product) against the previous library: GEMM + bf16-mode tests on the product, pipeline A/B fp32 / bf16.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c41; mkdir -p $O; cd $R
V=$R/tools/micro/libsopro_prev.so
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_mode.py tests/test_gpu_full_size.py -q -x --timeout 200 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in prev new prev new; do
  L=""; [ $v = prev ] && L=$V
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q >> $O/f32_$v.json 2>> $O/f32_$v.err
done
for v in prev new prev new; do
  L=""; [ $v = prev ] && L=$V
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_$v.json 2>> $O/bf16_$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c41'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
grep -i "error\|Traceback" $O/*.err | head
