cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "skinny" 2>&1 | tail -2
for p in 1 0 1 0; do
  echo "pair=$p"
  SOPRO_SKINNY_PAIR=$p python tools/ar_concurrency_probe.py 32 400 2>&1 | grep -E "^1 phase, whole chip|2 phases, one shared 64|^1 phase, 64-CU"
  SOPRO_SKINNY_PAIR=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ttfa-runs 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-form', d['value'], d['ms_per_step'], d['phases_ms_per_step'] if 'phases_ms_per_step' in d else '')"
done
