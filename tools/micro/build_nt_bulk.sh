#!/bin/bash
# Developer A/B of the non-temporal policy for the throughput phases' big streams (csrc/common.h, SOPRO_NT_BULK: the product is 1 =
# always): builds the library with policy $1 (0 = never, 2 = only tensors >= 64 MB) as tools/micro/libsopro_nt_bulk$1.so (git-ignored); run with
# SOPRO_HIP_LIB=<that file> python bench.py ...
set -e
P=${1:-0}
cd "$(dirname "$0")/../../sopro_amd/csrc"
make -s
objs=$(ls *.o | grep -v -e gemm_bf16s.o -e seanet_res.o -e seanet_uptail.o -e gemm_f32.o)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DSOPRO_NT_BULK=$P"
for f in gemm_bf16s seanet_res seanet_uptail gemm_f32; do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/nt${P}_$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/libsopro_nt_bulk$P.so /tmp/nt${P}_gemm_bf16s.o /tmp/nt${P}_seanet_res.o /tmp/nt${P}_seanet_uptail.o /tmp/nt${P}_gemm_f32.o $objs
