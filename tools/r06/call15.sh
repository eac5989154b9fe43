#!/bin/bash
# r06 call 15: shader clock under each kernel family of the throughput partition (which ones are power-limited?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c15; mkdir -p $O; cd $R
timeout 600 python tools/r06/kernel_clock_probe.py > $O/kernel_clock.txt 2> $O/kernel_clock.err; echo rc $?; cat $O/kernel_clock.txt | cut -c1-220; tail -3 $O/kernel_clock.err
