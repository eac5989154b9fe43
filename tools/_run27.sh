cd /root/repo
for r in 1 2 3 4 5 6 7 8; do
  SOPRO_BENCH_TRACE=2 python bench.py --frames 400 --steps 12 --warmup 4 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2> gpurun_out/trace_$r.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
uptime
python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for r in 1 2 3 4; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-form', d['value'], d['ms_per_step'])"; done
