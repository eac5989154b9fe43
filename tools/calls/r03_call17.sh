#!/bin/bash
# round 3, call 17: SQ counters of the SEANet kernels (decoder alone, 32 x 200 frames)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/mimi_probe.py"
export PROBE_B=32
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq1 -o a -- $P > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -o b -- $P > $O/sq2.log 2>&1
cd $R
python - <<'P'
import csv, collections, glob
for d in ('sq1','sq2'):
    fs = glob.glob(f'gpurun_out/r03i/{d}/**/*counter_collection.csv', recursive=True)
    print(d, fs)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][-60:]
        if not any(t in k for t in ('seanet', 'attn_mfma')): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen:
            seen.add(r['Dispatch_Id']); n[k]+=1; acc[k]['ns'] += int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    for k, v in acc.items():
        print(k, 'launches', n[k], {c: round(x / n[k]) for c, x in v.items()})
P
rm -f $O/*/*/*_kernel_trace.csv $O/*/*_kernel_trace.csv
