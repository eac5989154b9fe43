#!/bin/bash
# r04 call 40: the contraction's global requests dealt out between the MFMAs of a K-step (-DSOPRO_GEMM_SPREAD) against the burst
# behind the barrier (product): GEMM tests on the variant, the loop probe on both, pipeline A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c40; mkdir -p $O; cd $R
V=$R/tools/micro/libsopro_gemm_spread.so
SOPRO_HIP_LIB=$V timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_mode.py -q -x -k "gemm" --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest (variant) rc $?"; tail -3 $O/pytest.log | cut -c1-300
for cus in "" 192; do
  PROBE_CUS=$cus timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  PROBE_CUS=$cus SOPRO_HIP_LIB=$V timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  echo "---- (above: ${cus:-256} CUs)" >> $O/loop.txt
done
grep -v amdgpu.ids $O/loop.txt
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for v in prod var prod var; do
  L=""; [ $v = var ] && L=$V
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q >> $O/f32_$v.json 2>> $O/f32_$v.err
done
for v in prod var; do
  L=""; [ $v = var ] && L=$V
  SOPRO_HIP_LIB=$L timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_$v.json 2>> $O/bf16_$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c40'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('ok'), d['parity'].get('timed_steps_identical'))
P
grep -i "error\|Traceback" $O/*.err | head
