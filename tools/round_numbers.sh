# Collects the side measurements DESIGN.md quotes (run on the GPU box from the repo root): bash tools/round_numbers.sh r02
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout -k 10 200 python bench.py --node-kind igsfa --no-cpu-baseline > gpurun_out/${TAG}_bench_igsfa.json 2>/dev/null
timeout -k 10 200 python bench.py --input-dtype uint8 --no-cpu-baseline --no-frame > gpurun_out/${TAG}_bench_u8.json 2>/dev/null
timeout -k 10 300 python tools/host_path.py 2>/dev/null > gpurun_out/${TAG}_host_path.txt
timeout -k 10 300 python tools/small_batch2.py 2>/dev/null | grep "^N=" > gpurun_out/${TAG}_small_batch.txt
timeout -k 10 300 python tools/prod_stage_times.py 128 2>/dev/null > gpurun_out/${TAG}_product_plan.txt
timeout -k 10 300 python tools/train_bench.py 2>/dev/null > gpurun_out/${TAG}_train_bench.txt
timeout -k 10 600 python tools/train_hier_bench.py 2>/dev/null > gpurun_out/${TAG}_train_hier.txt
echo done; tail -c 300 gpurun_out/${TAG}_train_hier.txt
