#!/bin/bash
# round 3, call 43: final build - the tests that the last two edits touch, then the round's bench lines
cd "$GRAFT_REPO_ROOT"
S=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_ops.py -x -q -m gpu -k "profiler or xattn or attention" 2>&1 | tail -2
echo "tests: $(( $(date +%s) - S )) s"; S=$(date +%s)
bash tools/collect_evidence.sh r03f bench 2>&1 | tail -12
echo "bench: $(( $(date +%s) - S )) s"
