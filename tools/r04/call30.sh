#!/bin/bash
# r04 call 30 (last of the round; the library with non-temporal output streams): the whole GPU suite + smoke on the final library, then the round's evidence (kernel stats, PMC
# passes, AR kernel table, bench lines) from the same library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c30; mkdir -p $O; cd $R
timeout 1300 python -m pytest tests -m gpu -q -x --timeout 200 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -6 $O/pytest_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
bash tools/collect_evidence.sh r04y all 2>&1 | tail -14 | cut -c1-500
