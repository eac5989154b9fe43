#!/bin/bash
# Developer timing builds of the fused last SEANet level (tools/micro/seanet_uptail_ablate.hip = csrc/seanet_uptail.hip + SOPRO_UPTAIL_ABL hooks): the library with one phase of
# the kernel left out (wrong results - what the remaining phases cost is the point).  Writes tools/micro/libsopro_uptail_abl<N>.so
# (git-ignored); run with SOPRO_HIP_LIB=<that file> python tools/uptail_probe.py 64 96000 fused
# bits: 1 transposed convolution (matrix-core part), 2 first convolution, 4 second convolution, 8 last layer, 16 ELU + split of h, 32 x staging
set -e
cd "$(dirname "$0")/../../sopro_amd/csrc"
make -s
objs=$(ls *.o | grep -v seanet_uptail.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DSOPRO_UPTAIL_ABL=$n -I. -c ../../tools/micro/seanet_uptail_ablate.hip -o /tmp/seanet_uptail_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/libsopro_uptail_abl$n.so /tmp/seanet_uptail_abl$n.o $objs
done
