cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1'.ljust(34), round(d['value']),round(d['ms_per_step'],4),[round(x*1e3,1) for x in d['roofline']['stages_ms'][:9]], d.get('max_rel_err_vs_oracle'))"; }
run default
HIGSFA_WHATIF_OVERLAP=3 run overlap_from_3
HIGSFA_WHATIF_OVERLAP=2 run overlap_from_2
HIGSFA_WHATIF_OVERLAP=4 run overlap_from_4
run default
