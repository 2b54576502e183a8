# Kernel traces of the default bench under launch-shape knobs for the top of the hierarchy (run on the GPU box):
#   bash tools/top_layers.sh <tag> "ENV=1 ENV2=2" ...     one trace per quoted environment setting ("" = defaults)
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ENVS in "$@"; do
  D=$R/gpurun_out/top_${TAG}_$i
  ( export $ENVS HIGSFA_DUMMY=1; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $D.log 2>&1 )
  echo "=== [$ENVS]" >> $R/gpurun_out/top_$TAG.txt
  python3 $R/tools/kernel_positions.py $D >> $R/gpurun_out/top_$TAG.txt
  tail -1 $D.log | cut -c1-220 >> $R/gpurun_out/top_$TAG.txt
  i=$((i+1))
done
