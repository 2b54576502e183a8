set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rows in 1024 2048 4096 8192; do python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --rows $rows 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows', d['config']['rows_per_gpu'], 'val %.0f'%d['value'], 'ms %.3f'%d['ms_per_step'], d['roofline']['stages_ms'])"; done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1 || tail -5 $R/gpurun_out/pmc1.log
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1 || tail -5 $R/gpurun_out/pmc2.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1 || tail -5 $R/gpurun_out/pmc3.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1 || tail -5 $R/gpurun_out/pmc4.log
ls $R/gpurun_out/pmc*/*/ | head -30
