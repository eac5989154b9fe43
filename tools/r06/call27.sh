#!/bin/bash
# r06 call 27: res128 with row tile 1's skip loads issued a row tile ahead (product library) against the previous form (developer library built before the change), alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c27; mkdir -p $O; cd $R
for i in 1 2 3; do
  timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64" | sed "s/^/new /"
  SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64" | sed "s/^/old /"
done
