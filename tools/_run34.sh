cd /root/repo
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_continuous.py tests/test_gpu_full_size.py tests/test_abi.py -x -q 2>&1 | tail -2
fails=0
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fl_$i.out 2> gpurun_out/fl_$i.err || fails=$((fails+1)); done
echo "f32 full-line failures: $fails of 10"
fails=0
for i in 11 12 13 14 15 16; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > gpurun_out/fl_$i.out 2> gpurun_out/fl_$i.err || fails=$((fails+1)); done
echo "bf16 full-line failures: $fails of 6"
