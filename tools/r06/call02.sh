#!/bin/bash
# r06 call 2: the 8-phase LDS-DMA contraction prototype (tools/micro/gemm8p_proto.hip) on the decoder's K >= 1024 shapes (64-row chunk),
# whole chip and the 192-CU partition, with / without the wave-group stagger; the product tile kernel on the same box beside it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c02; mkdir -p $O; cd $R/tools/micro
for v in "1 8" "0 8" "1 6"; do set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTAGGER=$1 -DVMWAIT=$2 -o /tmp/g8p_$1_$2 gemm8p_proto.hip || exit 1
done
{
for shape in "25600 4096 2048" "204800 1536 1024" "25600 512 2048" "25600 1024 3584" "204800 256 1536" "25600 1536 512"; do
  for cus in 256 192; do
    for v in 1_8 0_8 1_6; do timeout 300 /tmp/g8p_$v $shape $cus 20; done
  done
done
} 2>&1 | tee $O/gemm8p.txt
cd $R
PROBE_CUS=192 timeout 600 python tools/gemm_split_probe.py 1 2>&1 | grep -v amdgpu.ids | tee $O/product_192.txt | cut -c1-200
