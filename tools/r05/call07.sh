#!/bin/bash
# r05 call 7: the whole GPU suite + smoke on the scheduler changes (auto pass sizes, prepare, conditioning gate), then two kernel
# traces of the pipelined bench that differ only in the number of timed steps (tools/trace_delta.py: what runs PER TIMED STEP)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c07; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -16
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --warmup 5 --coalesce 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t8 -o t8 -- $B --steps 8 > $O/t8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t24 -o t24 -- $B --steps 24 > $O/t24.log 2>&1
rm -f $O/t8/*_kernel_trace.csv $O/t24/*_kernel_trace.csv
cd $R
python tools/trace_delta.py $O/t8/t8_kernel_stats.csv $O/t24/t24_kernel_stats.csv 16 $O/r05_timed_path_kernels.json
