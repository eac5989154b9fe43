#!/bin/bash
# r04 call 19: 128x128 contraction tiles on eight waves (SOPRO_GEMM_W8=0: the four-wave form): the equivalence test + the GEMM tests,
# pipeline A/B in fp32 and bf16 mode.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c19; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_mode.py -q -x -k "gemm" --timeout 150 --timeout-method=thread > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc $?"; tail -15 $O/pytest_gemm.log | cut -c1-500
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for w in 0 1 0 1; do
  SOPRO_GEMM_W8=$w timeout 300 python bench.py $Q >> $O/f32_w8_$w.json 2>> $O/f32_w8_$w.err
done
for w in 0 1 0 1; do
  SOPRO_GEMM_W8=$w timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_w8_$w.json 2>> $O/bf16_w8_$w.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c19'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'), d.get('output_hash') or d.get('parity',{}).get('hash'))
        except Exception as e: print(f, 'ERR', e)
P
grep -i "error\|Traceback" $O/*.err | head
