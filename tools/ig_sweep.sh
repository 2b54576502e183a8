# iGSFA workgroup-shape sweep (experiments): per-stage times for forced shapes vs the automatic choice
set -e
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --node-kind igsfa 2>gpurun_out/ig_dbg.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), d['max_rel_err_vs_oracle'], d['roofline']['stages_ms'])"; }
HIGSFA_DEBUG=1 run auto
sort -u gpurun_out/ig_dbg.txt | grep igsfa
for sh in "$@"; do HIGSFA_IG_SHAPE=$sh run $sh; done
