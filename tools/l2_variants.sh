cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-frame 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['roofline']['stages_ms'];print('$1: ms/step %.4f  L2 %.1f L3 %.1f L4 %.1f us' % (d['ms_per_step'], s[2]*1e3, s[3]*1e3, s[4]*1e3))"; }
run "rem4 + prefetch-all"
HIGSFA_NO_REM4=1 run "no rem4, prefetch-all"
HIGSFA_NO_PREFETCH_ALL=1 run "rem4, streaming"
HIGSFA_NO_REM4=1 HIGSFA_NO_PREFETCH_ALL=1 run "no rem4, streaming"
