#!/bin/bash
# r06 call 25: SQ counters of the decoder's kernels (what do the window attention and the fused SEANet kernels wait for?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c25; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | cut -c1-3000 > $O/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $set | cut -d' ' -f1)
  DECODE_EAGER=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o c -- python $R/tools/r06/decode_run.py 192 2 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'P'
import csv, sys, collections, re
if not sys.argv[1]: sys.exit()
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    if not any(k in name for k in ("attn_mfma_split", "seanet_uptail", "seanet_res128", "gemm_8p", "gemm_bf16s_kernel<2, 2, 2, 2, 2, 1,")): continue
    d[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(r["Dispatch_Id"]); 
    if (name,key) not in seen: seen.add((name,key)); n[name]+=1
for name in d:
    print(name, "dispatches", n[name], {k: f"{v / n[name]:.4g}" for k, v in d[name].items()})
P
  rm -f $f
done
