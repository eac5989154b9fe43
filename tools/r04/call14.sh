#!/bin/bash
# r04 call 14: fused last level after the LDS-traffic changes (4-byte stores through lane pairs, first-convolution fragments in
# registers for the one-pass forms, 2048 workgroups): unit tests, probe, and what each phase costs (ablation builds).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c14; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_uptail.py -q --timeout 120 --timeout-method=thread > $O/pytest_uptail.log 2>&1; echo "pytest uptail rc $?"; tail -25 $O/pytest_uptail.log | cut -c1-600
timeout 240 python tools/uptail_probe.py > $O/probe.txt 2>&1; echo "probe rc $?"; cat $O/probe.txt | tail -40
timeout 100 python tools/uptail_probe.py 64 96000 fused > $O/abl.txt 2>&1
for n in 1 2 4 8 16 30; do
  SOPRO_HIP_LIB=$R/tools/micro/libsopro_uptail_abl$n.so timeout 100 python tools/uptail_probe.py 64 96000 fused >> $O/abl.txt 2>&1
done
grep -v amdgpu.ids $O/abl.txt
