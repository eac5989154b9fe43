#!/bin/bash
# r06 call 4: which waves share a SIMD?  The stagger must split the two waves of each SIMD into different groups.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c04; mkdir -p $O; cd $R/tools/micro
for v in 0 1 2 3; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTAGGER=$v -o /tmp/g8p_s$v gemm8p_proto.hip || exit 1; done
{
for v in 0 1 2 3; do
  for k in 1024 4096; do timeout 120 /tmp/g8p_s$v 4096 4096 $k 256 20; done
  for k in 1024 4096; do timeout 120 /tmp/g8p_s$v 3072 4096 $k 192 20; done
done
} 2>&1 | tee $O/stagger.txt
