#!/bin/bash
# r06 call 40: f16 operand scales undone in the epilogue's multiply-add (+ row scales by 16-byte LDS reads): product library against the
# library of call 39's straight-line epilogue alone (libsopro_hip_prev.so); whole suite on product and developer libraries
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c40; mkdir -p $O; cd $R
P="SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_prev.so"; D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
for i in 1 2 3; do
  echo "prev:"; env $P timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "new:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
echo "prev:"; env $P timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"; echo "new:"; timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"
env $D timeout 600 python tools/r06/tile_life.py 2>&1 | grep " x " | tee $O/tile_life.txt | cut -c1-44,120-400
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu (product) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
env $D timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_dev.log 2>&1; echo "pytest gpu (developer library) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_dev.log | cut -c1-260 | tail -8
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  timeout 300 python bench.py $Q > $O/b_$i.json 2> $O/b_$i.err
  python - <<PY
import json
d=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
PY
done
