for combo in "tests/test_gpu_full_size.py" "tests/test_gpu_serving.py tests/test_gpu_stages.py" "tests/test_gpu_pipeline.py tests/test_gpu_stages.py" "tests/test_gpu_ops.py tests/test_gpu_stages.py" "tests/test_gpu_full_size.py tests/test_gpu_stages.py"; do
  echo "=== $combo"; timeout 600 python -m pytest $combo -q -p no:randomly 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | head -8
done
