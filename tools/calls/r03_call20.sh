#!/bin/bash
# round 3, call 20: resident FF1 -> FF2 pair against two graph launches (tools/micro/persist_ff_proto.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/persist_ff tools/micro/persist_ff_proto.hip 2>/dev/null || exit 1
timeout 120 /tmp/persist_ff 2>&1 | grep -v amdgpu.ids | tee $O/persist_ff.txt
