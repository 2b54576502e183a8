# Per-layer sweep of the k_stage workgroup shape and tile parts at N = 4096 (experiment knobs HIGSFA_STAGE_SHAPE / _PARTS / _ONLY):
#   bash tools/stage_shape_sweep.sh   -> us per call for every (stage, shape) and (stage, parts); first line: the planner's own choice
cd $GRAFT_REPO_ROOT
echo -n "planner            "; timeout -k 10 100 python tools/call_times.py 4096 2>/dev/null
for si in 2 3 4 5 6 7; do
  for sh in 4,2 8,1 4,1; do
    echo -n "stage $si shape $sh   "; HIGSFA_STAGE_ONLY=$si HIGSFA_STAGE_SHAPE=$sh timeout -k 10 100 python tools/call_times.py 4096 2>/dev/null
  done
  for pp in 2 4 6 8 12 16 100; do
    echo -n "stage $si parts $pp   "; HIGSFA_STAGE_ONLY=$si HIGSFA_STAGE_PARTS=$pp timeout -k 10 100 python tools/call_times.py 4096 2>/dev/null
  done
done
echo -n "planner            "; timeout -k 10 100 python tools/call_times.py 4096 2>/dev/null
