#!/bin/bash
# r06 call 13: would the long-K form pay on up2 (1228800 x 640, K = 512)?  Prototype at N = 768 (three full column tiles) and N = 512, K = 512, 192 CUs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c13; mkdir -p $O; cd $R/tools/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMAPOS=2 -DVMWAIT=6 -o /tmp/g8p gemm8p_proto.hip || exit 1
{
timeout 300 /tmp/g8p 614400 768 512 192 5
timeout 300 /tmp/g8p 614400 512 512 192 5
timeout 300 /tmp/g8p 614400 768 512 256 5
} 2>&1 | grep -v HW_ID | tee $O/up2.txt
