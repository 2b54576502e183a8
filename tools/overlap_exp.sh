# experiment: two half-batches on two streams, staggered, with grids sized for a share of the chip (bench.py wall numbers)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['value']),round(d['ms_per_step'],4),'err %.1e' % d['max_rel_err_vs_oracle'])"; }
run "baseline            "
for occ in 1.0 0.67 0.5; do for stg in -1 2 3 4; do
  HIGSFA_SPLIT=2 HIGSFA_OCC_SCALE=$occ HIGSFA_STAGGER=$stg run "split2 occ=$occ stagger=$stg"
done; done
