set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_cascade.py tests/test_grid_patches.py -x -q -m gpu > gpurun_out/r4_cascade_tests.log 2>&1 || { tail -40 gpurun_out/r4_cascade_tests.log; exit 1; }
tail -3 gpurun_out/r4_cascade_tests.log
timeout -k 10 600 python bench.py --no-extra-legs --no-cpu-baseline --no-inflight > gpurun_out/r4_bench_c.json 2> gpurun_out/r4_bench_c.err || { tail -20 gpurun_out/r4_bench_c.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_c.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('frame_leg'))[:900])
PY
