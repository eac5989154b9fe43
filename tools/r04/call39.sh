#!/bin/bash
# r04 call 39: where a wave of the contraction spends a K-step (timing build with shader-clock sums: tools/micro/gemm_kstep_stamps.patch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c39; mkdir -p $O; cd $R
SOPRO_HIP_LIB=$R/tools/micro/libsopro_gemm_stamps.so timeout 200 python tools/gemm_kstep_stamps.py > $O/stamps.txt 2>&1
echo "---- 192 CUs" >> $O/stamps.txt
PROBE_CUS=192 SOPRO_HIP_LIB=$R/tools/micro/libsopro_gemm_stamps.so timeout 200 python tools/gemm_kstep_stamps.py >> $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt
