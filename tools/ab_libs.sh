# Same-box A/B of prebuilt libraries (tools/build_ref_lib.sh): bash tools/ab_libs.sh [rounds] lib1 lib2 ...   ("-" = the in-tree build)
# Every round runs each library once, in order; prints value, ms/step and the per-stage event times.
cd $GRAFT_REPO_ROOT
N=${1:-2}; shift
run() { HIGSFA_LIB="$2" timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1'.ljust(34), round(d['value']),round(d['ms_per_step'],4),[round(x*1e3,1) for x in d['roofline']['stages_ms'][:9]])"; }
for i in $(seq $N); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then run "in-tree" ""; else run "$L" "$GRAFT_REPO_ROOT/$L"; fi
  done
done
