#!/bin/bash
# r04 call 11: whole GPU suite on the final binary + smoke.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c11; mkdir -p $O; cd $R
timeout 700 python -m pytest tests -m gpu -x -q --timeout 200 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
