#!/bin/bash
# round 3, call 48 (last GPU seconds): the split contraction's K loop as one basic block with pinned requests (SOPRO_ABLATE=16) against the product build
cd "$GRAFT_REPO_ROOT"
PROBE_REPS=4 timeout 30 python tools/gemm_ab_probe.py
SOPRO_HIP_LIB=tools/micro/libsopro_abl16.so PROBE_REPS=4 timeout 30 python tools/gemm_ab_probe.py
