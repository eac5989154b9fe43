O=gpurun_out/r03e; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_full_size.py 2>&1 | tail -40 ) > $O/pytest.log 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_full_size.py -q 2>&1 | tail -40 ) > $O/pytest_full.log 2>&1
grep -v "^$" $O/pytest.log | tail -30; grep -v "^$" $O/pytest_full.log | tail -30
