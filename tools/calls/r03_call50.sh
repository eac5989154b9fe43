#!/bin/bash
# round 3, call 50: every contraction test on the one-basic-block K loop (odd step counts, split-K, all piece counts)
cd "$GRAFT_REPO_ROOT"
timeout 28 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
