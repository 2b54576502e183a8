set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r5_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r5_gpu_tests.log
python bench.py --no-frame --no-extra-legs --no-cpu-baseline > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { tail -20 gpurun_out/r5_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('stages_ms'), d.get('batches_in_flight'))
if d.get('legs_failed'):
    raise SystemExit('bench legs failed: %s' % d['legs_failed'])
PY
