#!/bin/bash
# round 3, call 31: PMC passes again (the sixteen-wave tail kernel belongs to the tail family), batch 1 x 400 with four requests in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B --lanes 1 --steps 1 --warmup 2 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B --lanes 1 --steps 1 --warmup 2 > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $B --lanes 1 --steps 1 --warmup 2 > $O/mfma.log 2>&1
cd $R
python tools/pmc_summary.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/r03_pmc_summary.json | grep -A3 seanet_tail
python tools/pmc_summary.py --mfma $O/mfma/m_counter_collection.csv $O/r03_pmc_mfma_busy.json | grep -A8 seanet_tail
rm -f $O/*/*_kernel_trace.csv $O/*/*_counter_collection.csv
timeout 300 python bench.py --batch 1 --frames 400 --steps 16 --warmup 4 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2>/dev/null | tail -1 > $O/b1_400_l4.json
python -c "
import json
d=json.loads(open('$O/b1_400_l4.json').read()); print('1x400 x4 lanes', d['value'], d['ms_per_step'], d['config'].get('coalesce'))"
