#!/bin/bash
# r06 call 37: product library = residual pieces requested ahead + f16 staging folded into the MFMA stream: whole suite on the product and
# on the developer library, bench (driver's form) x 2, the 1 x 4-wave tile on the fused-RMSNorm / GLU forms (same bits?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c37; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu (product) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
env $D timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu_dev.log 2>&1; echo "pytest gpu (developer library) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu_dev.log | cut -c1-260 | tail -8
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2 3; do
  timeout 300 python bench.py $Q > $O/b_$i.json 2> $O/b_$i.err
  python - <<P
import json
d=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
done
echo "--- tiles 1 / 3 (developer library)"; env $D timeout 600 python tools/r06/tile_probe.py 1 3 2>&1 | grep " x " | tee $O/tile_probe.txt | cut -c1-200
