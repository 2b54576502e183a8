# experiment: launch shapes of the small top layers at N = 4096 (stage times from bench.py)
cd $GRAFT_REPO_ROOT
for cfg in "0 4" "1 4" "2 4" "1 0" "2 0" "1 2" "0 0"; do
  set -- $cfg
  HIGSFA_SHAPES=$1 HIGSFA_SPLITM_MAX=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-frame 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['roofline']['stages_ms'];print('shapes $1 splitm<=$2: ms/step %.4f  L5..L10 %s  sum6-10 %.1f us' % (d['ms_per_step'], s[5:11], sum(s[6:11])*1e3))"
done
