#!/bin/bash
# r04 call 36: the three-pass contraction with its W fragments through LDS once per workgroup (tile override 9) against the product
# (1) and the eight-wave form (7): equivalence test, the decoder's shapes alone on the chip and on the 192-CU partition.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c36; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "eight_wave or odd_k or split_k" --timeout 150 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-300
PROBE_NOELU=1 timeout 300 python tools/gemm_split_probe.py 1,9 > $O/whole.txt 2>&1; grep -v amdgpu.ids $O/whole.txt | cut -c1-150
PROBE_NOELU=1 PROBE_CUS=192 timeout 300 python tools/gemm_split_probe.py 1,9 > $O/part192.txt 2>&1; grep -v amdgpu.ids $O/part192.txt | cut -c1-150
