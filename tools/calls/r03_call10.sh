echo "=== graphs off"; SOPRO_NO_BULK_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_stages.py -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | head -8
echo "=== graphs on, traceback"; timeout 600 python -m pytest tests/test_gpu_full_size.py -q --tb=short 2>&1 | grep -E "^E |^tests/|passed|failed" | head -20
