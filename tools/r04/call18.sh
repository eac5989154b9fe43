#!/bin/bash
# r04 call 18: the decoder's contraction on EIGHT waves per 128x128 tile (four waves per SIMD at two workgroups per CU) against the
# four-wave form, whole chip and 192-CU partition (tile overrides 1 = product, 7 = 2x4 waves of 64x32, 8 = 4x2 waves of 32x64).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c18; mkdir -p $O; cd $R
PROBE_NOELU=1 timeout 300 python tools/gemm_split_probe.py 1,7,8 > $O/whole.txt 2>&1; grep -v amdgpu.ids $O/whole.txt | cut -c1-200
PROBE_NOELU=1 PROBE_CUS=192 timeout 300 python tools/gemm_split_probe.py 1,7,8 > $O/part192.txt 2>&1; grep -v amdgpu.ids $O/part192.txt | cut -c1-200
