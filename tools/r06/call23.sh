#!/bin/bash
# r06 call 23: the GPU suite three times in a row on the final library (flake screen), smoke() and the driver's form once
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c23; mkdir -p $O; cd $R
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu_$i.log 2>&1; echo "run $i rc $?"
  grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu_$i.log | cut -c1-200 | tail -5
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"; tail -c 600 $O/bench_line.json
uptime
