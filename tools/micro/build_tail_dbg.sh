#!/bin/bash
# Developer probe: the library with shader-clock stamps inside the fused SEANet tail kernel (SOPRO_TAIL_DBG in seanet_tail.hip).
# Writes tools/micro/libsopro_taildbg.so (git-ignored); run with SOPRO_HIP_LIB=<that file> python tools/tail_timeline.py
set -e
cd "$(dirname "$0")/../../sopro_amd/csrc"
make -s
objs=$(ls *.o | grep -v seanet_tail.o)
build() {  # name, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DSOPRO_TAIL_DBG $2 -c seanet_tail.hip -o /tmp/seanet_tail_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/libsopro_taildbg$1.so /tmp/seanet_tail_$1.o $objs
}
build "" ""
build _noexp "-DTAIL_EXP_OFF"        # ELU replaced by the identity (wrong results): what the activation's instructions cost
