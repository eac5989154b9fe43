#!/bin/bash
# r05 call 8: the activation-stationary short-K contraction (csrc/gemm_astat.hip): equivalence test, then the probe (whole chip /
# 192 CUs) against the tile kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c08; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "activation_stationary" --timeout 240 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -12
timeout 300 python tools/astat_probe.py 2>&1 | tee $O/probe_whole.txt | cut -c1-200
PROBE_CUS=192 timeout 300 python tools/astat_probe.py 2>&1 | tee $O/probe_192.txt | cut -c1-200
