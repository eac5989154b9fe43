#!/bin/bash
# round 3, call 27: SQ counters of the sixteen-wave tail kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/tail_probe.py 32"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq1 -o a -- $P > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -o b -- $P > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH --kernel-trace --output-format csv -d $O/sq3 -o c -- $P > $O/sq3.log 2>&1
cd $R
python - <<'P'
import csv, collections, glob
for d,f in (('sq1','a'),('sq2','b'),('sq3','c')):
    try:
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
        for r in csv.DictReader(open(f'gpurun_out/r03s/{d}/{f}_counter_collection.csv')):
            k = r['Kernel_Name']
            if 'seanet_tail' not in k: continue
            k='tail16' if 'tail16' in k else 'tail4'
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if (r['Dispatch_Id']) not in seen:
                seen.add(r['Dispatch_Id']); n[k]+=1; acc[k]['ns'] += int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        for k, v in acc.items(): print(d, k, 'launches', n[k], {c: round(x / n[k]) for c, x in v.items()})
    except Exception as e: print(d, 'failed', e)
P
rm -f $O/*/*_kernel_trace.csv
