#!/bin/bash
# r04 call 44 (the round's last): the library with the contractions' requests dealt out between the MFMAs - whole GPU suite + smoke;
# if green: kernel stats of the pipelined fp32 / bf16 lines (copied over profiles/r04_bench_*lanes4_kernel_stats.csv ON THE BOX, so
# that the bench lines replay and stamp them), then the bench lines.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c44; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 200 --timeout-method=thread > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc $rc"; tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
if [ $rc -ne 0 ]; then echo "tests failed: no evidence collected"; exit 1; fi
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4 -o l4 -- $B --steps 8 --warmup 5 > $O/l4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4b -o l4b -- $B --steps 8 --warmup 5 --precision bf16 > $O/l4b.log 2>&1
rm -f $O/*/*_kernel_trace.csv
cd $R
[ -s $O/l4/l4_kernel_stats.csv ] && cp $O/l4/l4_kernel_stats.csv profiles/r04_bench_lanes4_kernel_stats.csv
[ -s $O/l4b/l4b_kernel_stats.csv ] && cp $O/l4b/l4b_kernel_stats.csv profiles/r04_bench_bf16_lanes4_kernel_stats.csv
bash tools/collect_evidence.sh r04c44 bench 2>&1 | tail -9 | cut -c1-420
