#!/bin/bash
# Whole-pipeline sweep of bench.py (one line per configuration into gpurun_out/sweep_r02.log): hardware queue count of the HIP
# runtime, lanes, concurrent AR phases, CUs of the generation partition.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$(env "${envs[@]}" timeout 240 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --ttfa-runs 0 --profile-steps 2 "$@" 2>/dev/null | tail -1)
  echo "$name $(echo "$out" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); r=[d["roofline"]]+d["roofline_more"]; ar=[e for e in r if e["kernel"].startswith("AR frame")]
    print(d["value"], d["ms_per_step"], "ar_us", ar[0]["avg_launch_us"] if ar else None, d["phase_ms_per_step"])
except Exception as e: print("FAILED", e)')" >> gpurun_out/sweep_r02.log
}
if [ "$1" = "a" ]; then
run base X=1 --
run hwq2 GPU_MAX_HW_QUEUES=2 --
run hwq8 GPU_MAX_HW_QUEUES=8 --
run hwq16 GPU_MAX_HW_QUEUES=16 --
run lanes6_parts3_cu96 X=1 -- --lanes 6 --ar-parts 3 --ar-cus 96
run hwq16_lanes6_parts3_cu96 GPU_MAX_HW_QUEUES=16 -- --lanes 6 --ar-parts 3 --ar-cus 96
run lanes6_parts3_cu64 X=1 -- --lanes 6 --ar-parts 3 --ar-cus 64
run cu96 X=1 -- --ar-cus 96
run cu48 X=1 -- --ar-cus 48
run cu32 X=1 -- --ar-cus 32
run lanes3 X=1 -- --lanes 3
run lanes1 X=1 -- --lanes 1
run bulk2 X=1 -- --bulk-slots 2
run notshared X=1 -- --ar-shared 0
fi
cat gpurun_out/sweep_r02.log
