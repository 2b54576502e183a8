# Kernel trace of the bench's frame leg: bash tools/run_frame_trace.sh [tag]   (HIGSFA_CASCADE_NO_GROUPS=1 in the environment: one stage per launch pair)
set -e
TAG=${1:-r5}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_frt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_frt -- python3 $GRAFT_REPO_ROOT/tools/frame_trace.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_frt.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/${TAG}_frt.log
f=$(ls gpurun_out/${TAG}_frt/*/*kernel_trace.csv | head -1)
python tools/trace_frame_summary.py $f 10
rm -rf gpurun_out/${TAG}_frt
