# Timing experiment (HIGSFA_DIAG build: tools/build_diag_lib.sh -> tools/ab/libhigsfa_diag.so; results WRONG): k_stage without its weight copy into LDS (HIGSFA_WHATIF=4)
# beside the same build unchanged — what hiding the copy behind the first GEMM could gain at most.
cd $GRAFT_REPO_ROOT
export HIGSFA_LIB=$GRAFT_REPO_ROOT/tools/ab/libhigsfa_diag.so
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1'.ljust(24), round(d['ms_per_step'],4),[round(x*1e3,1) for x in d['roofline']['stages_ms'][:9]])"; }
run diag_build
HIGSFA_WHATIF=4 run no_weight_copy
run diag_build
HIGSFA_WHATIF=4 run no_weight_copy
