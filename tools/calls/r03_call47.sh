#!/bin/bash
# round 3, call 47: the other users of the pipeline with the drain stream (8 ranks on one GPU, the serving loop)
cd "$GRAFT_REPO_ROOT"
timeout 100 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_serving.py -x -q -m gpu 2>&1 | tail -2
