set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests_b.log 2>&1 || { tail -40 gpurun_out/r4_gpu_tests_b.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests_b.log
python bench.py --no-frame --no-extra-legs --no-cpu-baseline > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err || { tail -20 gpurun_out/r4_bench_b.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('stages_ms'), d.get('batches_in_flight'))
PY
