#!/bin/bash
# round 3, call 42: final build - whole GPU suite, then the round's profile evidence (rocprofv3 kernel stats, PMC passes, AR kernel table)
cd "$GRAFT_REPO_ROOT"
S=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "tests: $(( $(date +%s) - S )) s"; S=$(date +%s)
bash tools/collect_evidence.sh r03f profiles 2>&1 | tail -30
echo "profiles: $(( $(date +%s) - S )) s"
