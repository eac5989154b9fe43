#!/bin/bash
# r06 call 12: what the conditioning phase (1.1 ms per step on the throughput partition) consists of; ELU via med3 check (uptail test + decode time)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c12; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_uptail.py tests/test_gpu_ops.py -m gpu -q --timeout 240 -p no:cacheprovider -k "uptail or mimi or seanet or tail or res128" > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -3 $O/pytest_a.log | cut -c1-300
for i in 1 2 3; do timeout 200 python tools/r06/decode_run.py 192 8 2>&1 | grep "decode 64"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cond -o t -- python $R/tools/r06/cond_run.py 192 4 > $O/cond.log 2>&1
grep conditioning $O/cond.log
f=$(find $O/cond -name "*kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    print(f'{float(r["TotalDurationNs"]) / 7 / 1e3:9.1f} us per pass  x{int(r["Calls"]) // 7:4d}  {float(r["TotalDurationNs"]) / tot * 100:5.1f} %  {(m.group(1) + (m.group(2) or ""))[:90]}')
P
find $O/cond -name "*kernel_trace.csv" -delete
