# A/B of compile-time switches on ONE box (boxes differ by 2-4 %): bash tools/ab_build.sh "-DHG_A_PINGPONG=0" ["-D... second variant" ...]
# The default build runs first and last; every variant is a forced rebuild with HIGSFA_CXXFLAGS.
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1'.ljust(44), round(d['value']),round(d['ms_per_step'],4),[round(x*1e3,1) for x in d['roofline']['stages_ms'][:9]])"; }
run default; run default
for v in "$@"; do
  HIGSFA_CXXFLAGS="$v" timeout -k 10 900 python -m pyfaceanalysis_amd.build --force > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  run "[$v]"; run "[$v]"
done
timeout -k 10 900 python -m pyfaceanalysis_amd.build --force > gpurun_out/ab_build.log 2>&1
run default
