cd $GRAFT_REPO_ROOT
for s in 2 3 4; do
  HIGSFA_NO_REM4=1 HIGSFA_STAMP=$s timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame --no-inflight 2>&1 | grep "stamp stage" | tail -2
done
HIGSFA_STAMP=0 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame --no-inflight 2>&1 | grep "stamp stage" | tail -1
