#!/bin/bash
# Developer aid: a rocprofv3 kernel trace of the pipelined bench, reduced to (name, queue, stream, start, end) rows in
# gpurun_out/kt_compact.csv.gz (the full trace is ~20 MB): input of ad-hoc timeline analyses (slot occupancy, gaps).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 --steps 12 --warmup 6 "$@" > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-120
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" "$R" <<'P'
import csv, sys, gzip
rows = csv.DictReader(open(sys.argv[1]))
with gzip.open(sys.argv[2] + "/gpurun_out/kt_compact.csv.gz", "wt") as g:
    g.write("name,queue,stream,start,end\n")
    for r in rows:
        short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70].replace(",", ";")
        g.write(f'{short},{r.get("Queue_Id","")},{r.get("Stream_Id","")},{r["Start_Timestamp"]},{r["End_Timestamp"]}\n')
P
