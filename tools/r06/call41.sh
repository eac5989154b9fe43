#!/bin/bash
# r06 call 41: which forms of the tile kernel changed bits with the f16 scales folded into the epilogue's multiply-add
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c41; mkdir -p $O; cd $R
SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_prev.so timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/prev.txt
timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/new.txt
paste -d'|' $O/prev.txt $O/new.txt | awk -F'|' '{split($1,a,": "); split($2,b,": "); print a[1] ": " a[2] " " b[2] (a[2]==b[2] ? "" : "   <-- differs")}'
