#!/bin/bash
# r04 call 2: where does the first pipelined test hang (call 1: pytest timed out at test_two_lane_pipeline_equals_sequential)?
# pytest-timeout dumps every thread's stack; then the same file with the spin wait; then the new bf16-row kernel tests.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c02; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q --timeout 90 --timeout-method=thread > $O/pipe_wait1.log 2>&1; echo "rc $?" >> $O/pipe_wait1.log
tail -60 $O/pipe_wait1.log
SOPRO_BLOCKING_WAIT=0 timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q --timeout 90 --timeout-method=thread > $O/pipe_wait0.log 2>&1; echo "rc $?" >> $O/pipe_wait0.log
tail -5 $O/pipe_wait0.log
timeout 400 python -m pytest tests/test_gpu_bf16_mode.py -q --timeout 200 --timeout-method=thread > $O/bf16.log 2>&1; echo "rc $?" >> $O/bf16.log
tail -40 $O/bf16.log
