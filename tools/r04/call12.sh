#!/bin/bash
# r04 call 12: long-form legs with coalesced passes + chunked decode (A/B against the round-3 rule), the new chunk test.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c12; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -k "chunks or coalesced or two_lane" --timeout 200 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --frames 400 --steps 12 --warmup 6"
for lg in 0 1 0 1; do
  SOPRO_BENCH_COALESCE_LONG=$lg timeout 400 python bench.py $Q >> $O/f32_400_long$lg.json 2>> $O/f32_400_long$lg.err
done
for lg in 0 1; do
  SOPRO_BENCH_COALESCE_LONG=$lg timeout 400 python bench.py $Q --precision bf16 >> $O/bf16_400_long$lg.json 2>> $O/bf16_400_long$lg.err
done
python - <<'P'
import json,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04c12'
for f in sorted(glob.glob(O+'/*.json')):
    for l in open(f).read().strip().splitlines():
        try:
            d=json.loads(l)
            print(os.path.basename(f), d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['config']['coalesce'], d['parity'].get('timed_steps_identical'))
        except Exception as e: print(f, 'ERR', e)
P
grep -i "error\|Traceback" $O/*.err | head
