#!/bin/bash
# kernel + HIP API + copy trace of tools/rccl_world1.py (40 timed steps per variant): what sits between the last kernel of a step
# and the first kernel of the next one when the gather's hand-off is on.  No counters in this run.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r4_rccl_w1
rm -rf $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/rccl_world1.py 40 > $GRAFT_REPO_ROOT/gpurun_out/r4_rccl_w1.log 2>&1
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/r4_rccl_w1.log
ls $out/*/ | head
