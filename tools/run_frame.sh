# Cascade tests + the configs[2] frame leg, stage groups on and off (HIGSFA_CASCADE_NO_GROUPS) on one box: bash tools/run_frame.sh [tag]
set -e
TAG=${1:-r5}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_cascade.py tests/test_grid_patches.py -x -q -m gpu > gpurun_out/${TAG}_cascade_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_cascade_tests.log; exit 1; }
tail -3 gpurun_out/${TAG}_cascade_tests.log
for v in groups single groups single; do
  if [ $v = single ]; then export HIGSFA_CASCADE_NO_GROUPS=1; else unset HIGSFA_CASCADE_NO_GROUPS; fi
  timeout -k 10 600 python bench.py --no-extra-legs --no-cpu-baseline --no-inflight --steps 200 --warmup 50 > gpurun_out/${TAG}_bench_frame_$v.json 2> gpurun_out/${TAG}_bench_frame_$v.err || { tail -20 gpurun_out/${TAG}_bench_frame_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_frame_$v.json').read().strip().splitlines()[-1])
f=d.get('frame_leg') or {}
print('$v', 'ms/step', d['ms_per_step'], 'frame ms', f.get('ms_per_frame'), 'first stage', f.get('first_stage_ms'), 'rows', f.get('rows_executed'), f.get('survivors_per_stage'), {k: round(v['frames_per_s']) for k, v in (f.get('frames_in_flight') or {}).items()})
PY
done
