#!/bin/bash
# r06 call 53: which queues carry the __amd_rocclr_copyBuffer launches of a pipelined run (1639 in 13 steps, 11 us each: 1.5 % of kernel time)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06c53; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --no-legs --steps 8 --warmup 5 > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
print(rows[0].keys())
qk = "Queue_Id" if "Queue_Id" in rows[0] else [k for k in rows[0] if "ueue" in k][0]
fam = lambda n: "copyBuffer" if "copyBuffer" in n else ("skinny/xattn/sample" if ("skinny" in n or "xattn" in n or "ar_sample" in n) else ("gemm/seanet/attn" if ("gemm" in n or "seanet" in n or "attn" in n) else "other"))
c = collections.defaultdict(lambda: collections.Counter())
t = collections.defaultdict(float)
for r in rows:
    k = fam(r["Kernel_Name"])
    c[r[qk]][k] += 1
    if k == "copyBuffer": t[r[qk]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for q in sorted(c, key=lambda q: -sum(c[q].values())): print(q, dict(c[q]), "copy us", round(t[q], 1))
# sizes of the copies: grid sizes
g = collections.Counter((r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows if "copyBuffer" in r["Kernel_Name"])
print(g.most_common(12))
PY
rm -f $f
