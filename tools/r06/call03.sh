#!/bin/bash
# r06 call 3: what bounds the 8-phase loop?  One round of tiles (256 tiles on 256 CUs, 192 on 192), K = 1024 vs 4096: the difference is
# 96 K-tiles of pure loop.  Ablations: 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMAs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c03; mkdir -p $O; cd $R/tools/micro
for a in 0 1 2 3 4 8 9; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL=$a -o /tmp/g8p_a$a gemm8p_proto.hip || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL=0 -DSTAGGER=0 -o /tmp/g8p_ns gemm8p_proto.hip
{
for a in a0 ns a1 a2 a3 a4 a8 a9; do
  for k in 1024 4096; do timeout 120 /tmp/g8p_$a 4096 4096 $k 256 20; done
  for k in 1024 4096; do timeout 120 /tmp/g8p_$a 3072 4096 $k 192 20; done
done
} 2>&1 | tee $O/ablate.txt
