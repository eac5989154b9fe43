O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_serving.py "tests/test_gpu_ops.py::test_gemm_f16x3_is_fp32_class" -x -q 2>&1 | tail -60 > $O/a.log
timeout 600 python -m pytest tests/test_gpu_serving.py -q 2>&1 | grep -v "^$" | tail -60 > $O/b.log
timeout 600 python -m pytest "tests/test_gpu_ops.py::test_gemm_f16x3_is_fp32_class" -q 2>&1 | grep -E "^E |Error|assert" | head -40 > $O/c.log
cat $O/a.log; echo ======; cat $O/b.log; echo =====; cat $O/c.log
