#!/bin/bash
# round 3, call 15: MFMA cross-attention, library profiler, wide AR tiles in the pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03g; O=gpurun_out/r03g
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" ; tail -3 $O/pytest_a.log
timeout 200 python tools/attn_probe.py > $O/attn_probe.txt 2>&1; cat $O/attn_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
SOPRO_AR_TILES_WIDE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --ttfa-runs 0 > $O/bench_narrow.json 2> $O/bench_narrow.err; echo "narrow rc=$?"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
