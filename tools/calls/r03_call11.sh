echo "=== graphs on, no split-K"; SOPRO_NO_SPLITK=1 timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_stages.py -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | head -8
