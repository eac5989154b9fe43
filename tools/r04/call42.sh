#!/bin/bash
# r04 call 42: requests dealt out between the MFMAs, W fragment addresses as (K-step base) + (lane offset) + immediate
# (tools/micro/gemm_spread_requests_scalar_base.patch: every family) against the product: loop probe, GEMM tests on the variant.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c42; mkdir -p $O; cd $R
V=$R/tools/micro/libsopro_gemm_spread2.so
SOPRO_HIP_LIB=$V timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest (variant) rc $?"; tail -2 $O/pytest.log | cut -c1-200
for cus in "" 192; do
  PROBE_CUS=$cus timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  PROBE_CUS=$cus SOPRO_HIP_LIB=$V timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  echo "---- (above: ${cus:-256} CUs)" >> $O/loop.txt
done
grep -v amdgpu.ids $O/loop.txt
