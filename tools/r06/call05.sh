#!/bin/bash
# r06 call 5: per-span clock stamps of the 8-phase loop and the DMA placement variants
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c05; mkdir -p $O; cd $R/tools/micro
for v in "0 8" "1 8" "2 6"; do set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTAMPS=1 -DDMAPOS=$1 -DVMWAIT=$2 -o /tmp/g8p_sd$1 gemm8p_proto.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMAPOS=$1 -DVMWAIT=$2 -o /tmp/g8p_d$1 gemm8p_proto.hip || exit 1
done
for a in 1 2 3; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTAMPS=1 -DABL=$a -o /tmp/g8p_sa$a gemm8p_proto.hip || exit 1; done
{
for v in sd0 sd1 sd2 sa1 sa2 sa3; do timeout 120 /tmp/g8p_$v 3072 4096 4096 192 10; done
for v in d0 d1 d2; do
  for k in 1024 4096; do timeout 120 /tmp/g8p_$v 3072 4096 $k 192 20; done
  timeout 120 /tmp/g8p_$v 4096 4096 4096 256 20
done
} 2>&1 | grep -v HW_ID | tee $O/stamps.txt
