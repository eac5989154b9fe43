#!/bin/bash
# The round's measured evidence in one call (run on an MI355X box through gpurun from the repo root) - make it the LAST call of the
# round, on the final binary:
#   tools/collect_evidence.sh r05 profiles    -> rocprofv3 kernel stats (pipelined fp32 / bf16, sequential), PMC passes, AR kernel table
#   tools/collect_evidence.sh r05 bench       -> the bench lines (driver's form, default, bf16 mode, other shapes)
#   tools/collect_evidence.sh r05 all         -> both
# Results land in gpurun_out/<tag>/; copy what is to be kept into profiles/ (see profiles/README.md for the names).
TAG=${1:-r06}; WHAT=${2:-bench}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O
if [ "$WHAT" = profiles ] || [ "$WHAT" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4 -o l4 -- $B --steps 8 --warmup 5 > $O/l4.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l4b -o l4b -- $B --steps 8 --warmup 5 --precision bf16 > $O/l4b.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l1 -o l1 -- $B --lanes 1 --steps 4 --warmup 2 > $O/l1.log 2>&1
  # counters in passes of their own, with the kernel trace only (no --stats, no sys / runtime traces).  The AR frame under the
  # counters is the PIPELINE's frame: 64 rows (two coalesced jobs), 1 x 2 workgroups, non-temporal folded operands - run as one
  # sequential 64-utterance batch (the counter collection serialises kernels anyway).  PMC_* tell tools/pmc_summary.py what ran.
  export SOPRO_AR_TILES=1x2 PMC_ROWS=64 PMC_PRECISION=f32
  export PMC_COMMAND="SOPRO_AR_TILES=1x2 python bench.py --lanes 1 --batch 64 --steps 1 --warmup 2 --profile-steps 0 --no-cpu-baseline --ttfa-runs 0 --no-legs (3 passes of a 64 x 200 step: the pipeline's coalesced frame)"
  P="$B --lanes 1 --batch 64 --steps 1 --warmup 2"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $P > $O/fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $P > $O/write.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $P > $O/mfma.log 2>&1
  unset SOPRO_AR_TILES
  # round 5: the same counters on the PIPELINED run - four lanes, CU partitions, 128-row passes as the timed run
  # issues them.  (The counter collection serialises dispatches, so the two partitions do not actually overlap under it; what the pass
  # adds over the sequential one is the pipeline's own launch shapes: frames of the coalesced row count, partition-sized grids.)
  SEQ_COMMAND="$PMC_COMMAND"
  export PMC_ROWS=128 PMC_PRECISION=f32
  export PMC_COMMAND="python bench.py --steps 8 --warmup 5 --profile-steps 0 --no-cpu-baseline --ttfa-runs 0 --no-legs (pipelined: 4 lanes, 64 + 192 CUs, four jobs per pass)"
  PP="$B --steps 8 --warmup 5"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pfetch -o f -- $PP > $O/pfetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pwrite -o w -- $PP > $O/pwrite.log 2>&1
  cd $R
  python tools/pmc_summary.py $O/pfetch/f_counter_collection.csv $O/pwrite/w_counter_collection.csv $O/${TAG}_pmc_summary_pipelined.json | head -30
  export PMC_ROWS=64 PMC_COMMAND="$SEQ_COMMAND"
  python tools/pmc_summary.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/${TAG}_pmc_summary.json | head -60
  python tools/pmc_summary.py --mfma $O/mfma/m_counter_collection.csv $O/${TAG}_pmc_mfma_busy.json | head -40
  python tools/ar_kernel_table.py $O/l1/l1_kernel_stats.csv $O/${TAG}_ar_kernels.json | head -40
  # (when the table is copied to profiles/rNN_ar_kernels.json, set its "source" to the committed name of the CSV it was made from:
  #  profiles/rNN_bench_lanes1_kernel_stats.csv - bench.py stamps that file's hash into the line)
  rm -f $O/*/*_kernel_trace.csv $O/*/*_counter_collection.csv  # the traces are large: keep the stats only
  ls $O/*/
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  cd $R
  # (round 6: stdout = the compact line the driver parses, < 8 kB; the full record is bench_full.json next to bench.py, kept per run)
  timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; cp bench_full.json $O/bench_line_full.json
  timeout 500 python bench.py > $O/bench_line_default.json 2> $O/bench_line_default.err; cp bench_full.json $O/bench_line_default_full.json
  timeout 300 python bench.py --steps 20 --warmup 5 --precision bf16 > $O/bench_line_bf16.json 2> $O/bench_line_bf16.err; cp bench_full.json $O/bench_line_bf16_full.json
  Q="--no-cpu-baseline --ttfa-runs 0 --profile-steps 0"
  ( echo '{"runs": [';
    timeout 300 python bench.py --frames 400 --steps 12 --warmup 4 $Q 2>/dev/null | tail -1; echo ',';
    timeout 300 python bench.py --batch 1 --frames 400 --lanes 1 --steps 12 --warmup 3 $Q 2>/dev/null | tail -1; echo ',';
    timeout 300 python bench.py --batch 1 --frames 400 --steps 16 --warmup 4 $Q 2>/dev/null | tail -1; echo ',';
    timeout 300 python bench.py --lanes 1 --steps 8 --warmup 2 $Q 2>/dev/null | tail -1;
    echo ']}' ) > $O/bench_other_shapes.json
  for f in bench_line bench_line_default bench_line_bf16; do python -c "
import json
t=open('$O/$f.json').read().strip().splitlines()[-1]
d=json.loads(t); r=d['roofline']
print('$f', len(t), 'bytes |', d['value'], d['ms_per_step'], d['phase_ms_per_step'], 'ttfa', d['config']['second_metric']['ttfa_ms_p50'], '| roofline', r['kernel'][:28], r['achieved'], r['frac'], 'us', r['avg_launch_us'], '| more', [(e['kernel'][:14], e['ms_per_step'], e.get('frac_of_pass_ceiling')) for e in d['roofline_more']], '| parity ok', d['parity'].get('ok'), d['parity']['timed_steps_identical'], '| cpu', (d.get('cpu_baseline') or {}).get('value'), '| legs', {k: v.get('value') for k, v in (d['config'].get('legs') or {}).items() if isinstance(v, dict)})"; done
  python -c "
import json
d=json.load(open('$O/bench_other_shapes.json'))
for r in d['runs']: print(r['config']['batch_per_gpu'], r['config']['frames'], r['config']['lanes_per_gpu'], r['value'], r['ms_per_step'], r['phase_ms_per_step'])"
  uptime
fi
