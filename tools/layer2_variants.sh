for v in "" "HIGSFA_NO_PACK=1" "HIGSFA_NO_PREFETCH_ALL=1" "HIGSFA_NO_REM4=1"; do
  env $v python bench.py --no-frame --no-extra-legs --no-cpu-baseline --no-inflight --steps 300 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(26), 'ms/step %.4f'%d['ms_per_step'], 'stages us', [round(x*1e3,1) for x in d['roofline']['stages_ms'][:9]])"
done
