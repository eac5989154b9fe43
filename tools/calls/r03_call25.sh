#!/bin/bash
# round 3, call 25: smaller generation partitions with fatter AR workgroups (the 64-CU floor came from 192-workgroup grids)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03q; O=gpurun_out/r03q
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
for cfg in "64 1x2" "48 1x2" "48 2x2" "32 2x2" "32 1x2" "64 2x2" "96 1x2"; do set -- $cfg
  SOPRO_AR_TILES_WIDE=$2 timeout 300 python bench.py $Q --ar-cus $1 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - "$O/b_$1_$2.json" "$cfg" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'])
except Exception as e: print(sys.argv[2], 'failed', e)
P
done | tee $O/sweep.txt
