#!/bin/bash
# r04 call 15: fused last level, second form (three phases; the transposed convolution's MFMAs dealt out between the vector
# instructions of the other stages; lane exchanges without LDS): unit tests, probe, ablations, pipeline A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c15; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_uptail.py -q --timeout 120 --timeout-method=thread > $O/pytest_uptail.log 2>&1; echo "pytest uptail rc $?"; tail -25 $O/pytest_uptail.log | cut -c1-700
timeout 240 python tools/uptail_probe.py > $O/probe.txt 2>&1; echo "probe rc $?"; cat $O/probe.txt | tail -40
timeout 100 python tools/uptail_probe.py 64 96000 fused > $O/abl.txt 2>&1
for n in 1 2 4 8 16; do
  SOPRO_HIP_LIB=$R/tools/micro/libsopro_uptail_abl$n.so timeout 100 python tools/uptail_probe.py 64 96000 fused >> $O/abl.txt 2>&1
done
grep -v amdgpu.ids $O/abl.txt
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 20 --warmup 5"
for fz in 0 1 0 1; do
  SOPRO_SEANET_FUSE=$fz timeout 300 python bench.py $Q >> $O/f32_fuse$fz.json 2>> $O/f32_fuse$fz.err
done
for fz in 0 1; do
  SOPRO_SEANET_FUSE=$fz timeout 300 python bench.py $Q --precision bf16 >> $O/bf16_fuse$fz.json 2>> $O/bf16_fuse$fz.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c15'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('ok'), d['parity'].get('timed_steps_identical'))
        except Exception as e: print(f, 'ERR', e)
P
grep -i "error\|Traceback" $O/*.err | head
