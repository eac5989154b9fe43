#!/bin/bash
# r04 call 38: what the K loop's workgroup barriers cost: the product library against a timing build without them
# (tools/micro/gemm_nosync.patch; wrong results), three-pass and one-pass contractions, whole chip and 192-CU partition.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c38; mkdir -p $O; cd $R
for cus in "" 192; do
  PROBE_CUS=$cus timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  PROBE_CUS=$cus SOPRO_HIP_LIB=$R/tools/micro/libsopro_gemm_nosync.so timeout 200 python tools/gemm_loop_probe.py >> $O/loop.txt 2>&1
  echo "---- (above: ${cus:-256} CUs)" >> $O/loop.txt
done
grep -v amdgpu.ids $O/loop.txt
