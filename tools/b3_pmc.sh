# EXPERIMENT: SQ counters of the split-bf16 stage kernel (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp HIGSFA_BF16X3=1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_b3a -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_b3a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc_b3b -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_b3b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_b3a","pmc_b3b"):
    f=max(glob.glob("$R/gpurun_out/%s/*/*_counter_collection.csv"%d))
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_stage_b3" not in r["Kernel_Name"]: continue
        key=(int(r["Dispatch_Id"]), r["Grid_Size"])
        c=per.setdefault(key, collections.Counter()); c[r["Counter_Name"]]+=float(r["Counter_Value"]); c["us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in list(per.items())[-5:]:
        print(d, k[1], {n:(round(x,1) if n=="us" else int(x)) for n,x in v.items()})
PY
