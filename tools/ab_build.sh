# A/B of a compile-time switch on one box: bash tools/ab_build.sh "-DHG_BRANCHY_EXPANSION=0"
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-inflight 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['value']),round(d['ms_per_step'],4),d['roofline']['stages_ms'])"; }
run default; run default
HIGSFA_CXXFLAGS="$1" timeout -k 10 600 python -m pyfaceanalysis_amd.build --force > gpurun_out/ab_build.log 2>&1
run "[$1]"; run "[$1]"
