# (each pass under `timeout`: a rocprofv3 that aborts can otherwise sit until the box's silence limit)
# Round profile: kernel stats of the default bench + HBM counters of the same command (separate
# passes, as MI355X_MICROARCH.md prescribes).  Run on the GPU box from the repo root:  bash tools/profile.sh r01
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/prof_$TAG.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_write_$TAG.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_sq_$TAG.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lds_$TAG -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_lds_$TAG.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mix_$TAG -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_mix_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-300
