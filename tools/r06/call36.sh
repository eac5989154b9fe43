#!/bin/bash
# r06 call 36: A = product library (evidence library of the round); V1 = developer library: residual pieces requested ahead + f16 staging folded
# into the MFMA stream; V2 = V1 without the f16 saturating clamp; + the 1 x 4-wave tile (override 3) for the f16 family / for every family
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c36; mkdir -p $O; cd $R
V1="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"; V2="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_v2.so"
nar() { env "$@" timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement; }
dec() { env "$@" timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"; }
for i in 1 2; do
  echo "A:"; nar X=1; dec X=1
  echo "V1:"; nar $V1; dec $V1
  echo "V2:"; nar $V2
  echo "V1 + f16 tile 3:"; nar $V1 SOPRO_F16X3_TILE=3
  echo "V2 + f16 tile 3:"; nar $V2 SOPRO_F16X3_TILE=3
  echo "V1 + every tile 3:"; dec $V1 SOPRO_GEMM_TILE=3
done
echo "--- V1 tile life"; env $V1 timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life_V1.txt | cut -c1-44,120-400
echo "--- V1 tiles 1 / 3"; env $V1 timeout 600 python tools/r06/tile_probe.py 1 3 2>&1 | grep " x " | tee $O/tile_probe_V1.txt | cut -c1-200
echo "--- V2 tiles 1 / 3"; env $V2 timeout 600 python tools/r06/tile_probe.py 1 3 2>&1 | grep "f16x3" | tee $O/tile_probe_V2.txt | cut -c1-200
env $V1 timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_v1.log 2>&1; echo "pytest (V1) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_v1.log | cut -c1-260 | tail -12
