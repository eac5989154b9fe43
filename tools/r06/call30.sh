#!/bin/bash
# r06 call 30: f16 three-pass staging without the saturating clamp (overflow -> inf, still counted by the range guard): product library built
# WITH the clamp (-DSOPRO_F16_CLAMP) against the developer library without it, same box, alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c30; mkdir -p $O; cd $R
for i in 1 2; do
  echo "--- clamp (run $i)"; timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep "f16x3" | tee -a $O/ff_clamp.txt | cut -c1-200
  echo "--- no clamp (run $i)"; SOPRO_DEV=1 timeout 600 python tools/r06/ff_cost_probe.py 2>&1 | grep "f16x3" | tee -a $O/ff_noclamp.txt | cut -c1-200
done
for i in 1 2; do
  echo "clamp:"; timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
  echo "no clamp:"; SOPRO_DEV=1 timeout 300 python tools/r06/nar_run.py 192 8 2>&1 | grep refinement
done
SOPRO_DEV=1 timeout 900 python -m pytest tests/test_gpu_range.py tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -m gpu -q --timeout 300 -p no:cacheprovider > $O/pytest_b.log 2>&1; echo "pytest (no clamp) rc $?"; tail -5 $O/pytest_b.log | cut -c1-300
