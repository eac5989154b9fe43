#!/bin/bash
# r04 call 5: micro-probes - the resident FF1 -> FF2 prototype at batch 1 / 4 / 16 / 32 rows, the bare MFMA loop on
# zero / random operands (item 4, first step), and the pipeline's lane trace after the refinement -> decode fusion.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r04c05; mkdir -p $O; cd $R
for b in r32_b16 r16_b16 r16_b4 r16_b1; do timeout 120 tools/micro/persist_ff_$b >> $O/persist_pair.txt 2>&1; done
cat $O/persist_pair.txt
timeout 120 tools/micro/mfma_peak > $O/mfma_peak.txt 2>&1; cat $O/mfma_peak.txt
SOPRO_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 > $O/f32_trace.json 2> $O/f32_trace.err
grep -E "idle|step  *[0-9]+ lane" $O/f32_trace.err | tail -30
