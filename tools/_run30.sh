cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "seanet" 2>&1 | tail -3
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_stages.py tests/test_gpu_full_size.py tests/test_gpu_bf16_mode.py -x -q 2>&1 | tail -3
for g in 1 0; do
  SOPRO_GEMM_UP=$g python bench.py --lanes 1 --steps 6 --warmup 3 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_up=$g seq', d['value'], d['ms_per_step'], d['phase_ms_per_step'])"
  for r in 1 2 3; do
  SOPRO_GEMM_UP=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_up=$g driver-form', d['value'], d['ms_per_step'], d['phase_ms_per_step'])"
  done
done
