#!/bin/bash
# r04 call 27: non-temporal policy for the throughput phases' big streams: 0 = never (the library of the earlier calls), 1 = always,
# 2 = by size (>= 64 MB; the product).  The unit tests that touch the changed stores, then pipeline A/B/C in fp32 and bf16.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c27; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_uptail.py tests/test_gpu_bf16_mode.py -q -x -k "gemm or seanet or uptail" --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-300
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
lib() { case $1 in 2) echo "";; *) echo $R/tools/micro/libsopro_nt_bulk$1.so;; esac; }
for v in 0 2 1 0 2 1; do
  SOPRO_HIP_LIB=$(lib $v) timeout 300 python bench.py $Q >> $O/f32_nt$v.json 2>> $O/f32_nt$v.err
done
for v in 0 2 1 0 2; do
  SOPRO_HIP_LIB=$(lib $v) timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_nt$v.json 2>> $O/bf16_nt$v.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c27'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        d=json.loads(l)
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('timed_steps_identical'))
P
grep -i "error\|Traceback" $O/*.err | head
