#!/bin/bash
# r06 call 55: the arg-max tests (with the new order-free reduction test) on the product library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 240 -p no:cacheprovider -k "argmax" 2>&1 | tail -5 | cut -c1-200
