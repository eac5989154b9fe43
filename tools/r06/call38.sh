#!/bin/bash
# r06 call 38: whole GPU suite on the product library, then the bench lines of the evidence set
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c38; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_evidence.sh r06 bench
