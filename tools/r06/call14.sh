#!/bin/bash
# r06 call 14: decoder chunk size: 128 x 200 frames in ONE chunk (SOPRO_MIMI_CHUNK_CELLS=25600) against two 64-row chunks, driver's form, alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c14; mkdir -p $O; cd $R
Q="--no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --warmup 5 --steps 20"
for i in 1 2; do
  for v in c12800 c25600 c6400; do
    SOPRO_MIMI_CHUNK_CELLS=${v#c} timeout 300 python bench.py $Q > $O/${v}_$i.json 2> $O/${v}_$i.err
    python - <<P
import json
d=json.loads(open('$O/${v}_$i.json').read().strip().splitlines()[-1])
print('%-8s' % '${v}', d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['parity'].get('rank_output_sha16'))
P
    grep -h "peak device memory" $O/${v}_$i.err | cut -c1-160
  done
done
