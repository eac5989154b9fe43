#!/bin/bash
# r06 call 51: GLU gate by the hardware exponential / reciprocal in the f16 three-pass and one-pass epilogues (product library) against the
# precise sigmoid (developer library: every-lane GLU epilogue of call 50): which forms change bits, refinement pass x 3, whole suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c51; mkdir -p $O; cd $R
D="SOPRO_DEV=1 SOPRO_HIP_LIB=$R/sopro_amd/libsopro_hip_dev.so"
timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/prod.txt
env $D timeout 600 python tools/r06/form_hash.py 2>&1 | grep " x " > $O/dev.txt
paste -d'|' $O/dev.txt $O/prod.txt | awk -F'|' '{split($1,a,": "); split($2,b,": "); print a[1] ": " a[2] " " b[2] (a[2]==b[2] ? "" : "   <-- differs")}' | tee $O/form_hash.txt | grep differs
for i in 1 2 3; do
  echo "precise:"; env $D timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "fast:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu (product) rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -8
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  timeout 300 python bench.py $Q > $O/b_$i.json 2> $O/b_$i.err
  python - <<PY
import json
d=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('ok'), d['parity'].get('rank_output_sha16'))
PY
done
