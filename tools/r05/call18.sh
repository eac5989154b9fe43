#!/bin/bash
# r05 call 18 (evidence, final library incl. the fused LayerNorms): whole GPU suite + smoke; if green: kernel stats / PMC passes (sequential AND pipelined), copied
# into profiles/r05_* ON THE BOX so that the bench lines replay and stamp them, then the bench lines.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r05c18; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail 12 --timeout 240 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc $rc"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-260 | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log | cut -c1-200
if [ $rc -ne 0 ]; then echo "tests failed: no evidence collected"; exit 1; fi
bash tools/collect_evidence.sh r05c18 profiles 2>&1 | tail -40 | cut -c1-260
for f in l4/l4_kernel_stats.csv:bench_lanes4_kernel_stats.csv l4b/l4b_kernel_stats.csv:bench_bf16_lanes4_kernel_stats.csv l1/l1_kernel_stats.csv:bench_lanes1_kernel_stats.csv; do
  [ -s $O/${f%%:*} ] && cp $O/${f%%:*} profiles/r05_${f##*:}
done
for j in pmc_summary pmc_mfma_busy pmc_summary_pipelined ar_kernels; do [ -s $O/r05c18_$j.json ] && cp $O/r05c18_$j.json profiles/r05_$j.json; done
python - <<'P'
import json
p='profiles/r05_ar_kernels.json'
try:
    d=json.load(open(p)); d['source']='profiles/r05_bench_lanes1_kernel_stats.csv'; json.dump(d,open(p,'w'),indent=1)
except Exception as e: print('ar_kernels', e)
P
mkdir -p $O/profiles_r05 && cp profiles/r05_bench_*kernel_stats.csv profiles/r05_pmc_*.json profiles/r05_ar_kernels.json $O/profiles_r05/ 2>/dev/null
bash tools/collect_evidence.sh r05c18 bench 2>&1 | tail -9 | cut -c1-520
