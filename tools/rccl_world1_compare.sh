# VERDICT r4 item 3 on ONE box: (1) the tool loop (RCCL in-process at world 1) with every variant, (2) bench.py under torch.distributed.run
# with one rank and --compare-collective (collective-free / side-stream / same-stream in one invocation), (3) plain bench.py.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== tools/rccl_world1.py 400 free,ordinary,light,same,free" > gpurun_out/r5_rccl_world1.txt
timeout -k 10 300 python tools/rccl_world1.py 400 free,ordinary,light,same,free 2>/dev/null >> gpurun_out/r5_rccl_world1.txt
echo "== python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --compare-collective (device-scope event if verified)" >> gpurun_out/r5_rccl_world1.txt
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 400 --warmup 100 --compare-collective --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2> gpurun_out/r5_bench_dist1.err | tail -1 > gpurun_out/r5_bench_dist1.json
python - >> gpurun_out/r5_rccl_world1.txt <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_dist1.json').read())
print('headline ms/step', round(d['ms_per_step'],4), d['config'].get('gather_events'))
print('stages_ms', d['roofline']['stages_ms'][:9])
print('collective_compare', d.get('collective_compare'))
PY
echo "== the same with --gather-stream same as the headline" >> gpurun_out/r5_rccl_world1.txt
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 400 --warmup 100 --gather-stream same --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>> gpurun_out/r5_bench_dist1.err | tail -1 > gpurun_out/r5_bench_dist1_same.json
python - >> gpurun_out/r5_rccl_world1.txt <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_dist1_same.json').read())
print('headline ms/step', round(d['ms_per_step'],4), d['config'].get('gather_events'))
print('stages_ms', d['roofline']['stages_ms'][:9])
PY
echo "== plain python bench.py (collective-free, no process group)" >> gpurun_out/r5_rccl_world1.txt
timeout -k 10 300 python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs 2>/dev/null | tail -1 > gpurun_out/r5_bench_plain.json
python - >> gpurun_out/r5_rccl_world1.txt <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_plain.json').read())
print('headline ms/step', round(d['ms_per_step'],4))
print('stages_ms', d['roofline']['stages_ms'][:9])
PY
cat gpurun_out/r5_rccl_world1.txt
