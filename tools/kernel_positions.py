"""Per-launch-position average kernel durations of a bench.py run traced with `rocprofv3 --kernel-trace --output-format csv`:
    python tools/kernel_positions.py gpurun_out/<dir> [skip_steps]
A step starts with the front kernel (k_stage0* / k_stage01*); the first `skip_steps` steps are dropped (warm-up)."""
import collections, csv, glob, os, sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
trace = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("hg::fused::(anonymous namespace)::", "").replace("hg::(anonymous namespace)::", "").replace("hg::fused::", "").replace("(TailParams)", "").replace("(StageParams, StageParams)", "").replace("(StageParams)", "").replace("(StageParams, int, int)", "")
per, gaps = collections.OrderedDict(), collections.OrderedDict()
pos, step, prev_end = -1, -1, None
for r in rows:
    k = short(r["Kernel_Name"])
    if not k.startswith("void k_"):
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k.startswith("void k_stage0"):
        pos, step = 0, step + 1
    else:
        pos += 1
    if step >= skip:
        per.setdefault((pos, k, r["Grid_Size_X"], r["Workgroup_Size_X"]), []).append((e - s) / 1e3)
        if prev_end is not None and pos > 0:
            gaps.setdefault(pos, []).append((s - prev_end) / 1e3)
    prev_end = e
tot = 0.0
for (p, k, g, wg), v in per.items():
    v.sort()
    gp = gaps.get(p, [0.0])
    print("%2d  %-62s grid %8s wg %4s  n %4d  avg %8.2f  med %8.2f us   gap before %5.2f" % (p, k[:62], g, wg, len(v), sum(v) / len(v), v[len(v) // 2], sum(gp) / len(gp)))
    tot += sum(v) / len(v)
print("sum of averages: %.1f us; top (positions >= 5): %.1f us" % (tot, sum(sum(v) / len(v) for (p, *_), v in per.items() if p >= 5)))
