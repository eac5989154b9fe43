#!/bin/bash
# round 3, call 21: non-temporal streams in the SEANet kernels, A/B (SOPRO_NT=1 built here, =0 rebuilt on the box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03m; O=gpurun_out/r03m
Q="--steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0"
run() { timeout 600 python bench.py $Q > $O/$1.json 2> $O/$1.err; python - "$O/$1.json" "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=[e for e in [d['roofline']]+d['roofline_more']]
print(sys.argv[2], d['value'], d['ms_per_step'], d['phase_ms_per_step'], {e['kernel'][:12]: e.get('avg_launch_us') for e in r})
P
}
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "seanet or tail or res128 or up128 or codec" 2>&1 | tail -1
PROBE_B=32 timeout 200 python tools/mimi_probe.py 2>&1 | grep -v "amdgpu.ids\|Exception\|Traceback\|hip.py\|Attribute" | tee $O/mimi_nt1.txt
run nt1_a; run nt1_b
cd sopro_amd/csrc && rm -f seanet_up.o seanet_res.o && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -DSOPRO_NT=0" > /dev/null 2>&1; cd "$GRAFT_REPO_ROOT"
PROBE_B=32 timeout 200 python tools/mimi_probe.py 2>&1 | grep -v "amdgpu.ids\|Exception\|Traceback\|hip.py\|Attribute" | tee $O/mimi_nt0.txt
run nt0_a; run nt0_b
