#!/bin/bash
# Developer probe: ablation builds of the split GEMM (see SOPRO_ABLATE in tools/micro/gemm_bf16s_ablate.hip, the developer copy of the kernel that carries the switches).
# Writes tools/micro/libsopro_abl<N>.so (git-ignored); run with SOPRO_HIP_LIB=<that file> python tools/gemm_split_probe.py 1
set -e
cd "$(dirname "$0")/../../sopro_amd/csrc"
make -s
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DSOPRO_ABLATE=$n -I. -c ../../tools/micro/gemm_bf16s_ablate.hip -o /tmp/gemm_bf16s_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/libsopro_abl$n.so /tmp/gemm_bf16s_abl$n.o \
    $(ls *.o | grep -v '^gemm_bf16s.o$')
done
