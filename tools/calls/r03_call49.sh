#!/bin/bash
# round 3, call 49: one-basic-block K loop without (16) / with (48) pinned requests against the product build
cd "$GRAFT_REPO_ROOT"
for lib in "" tools/micro/libsopro_abl16.so tools/micro/libsopro_abl48.so ""; do
  if [ -z "$lib" ]; then PROBE_REPS=4 timeout 20 python tools/gemm_ab_probe.py; else SOPRO_HIP_LIB=$lib PROBE_REPS=4 timeout 20 python tools/gemm_ab_probe.py; fi
done
