#!/bin/bash
# r04 call 20: the whole GPU suite and the smoke run on the round's last build (fused last SEANet level on by default).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c20; mkdir -p $O; cd $R
timeout 1300 python -m pytest tests -m gpu -q -x --timeout 200 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -12 $O/pytest_gpu.log | cut -c1-400
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log
