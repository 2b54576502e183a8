set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4_frt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_frt -- python3 $GRAFT_REPO_ROOT/tools/frame_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r4_frt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r4_frt/*/*kernel_trace.csv | head -1)
python tools/trace_frame_summary.py $f
