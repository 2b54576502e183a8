"""Summarise a rocprofv3 kernel trace of tools/frame_trace.py: the LAST `frames` frames (a frame starts with the prescale's table
kernel: the first k_extent_tables after a k_cascade_group / k_cascade_stage), per kernel name: launches and microseconds per frame."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
names = [short(r["Kernel_Name"]) for r in rows]
# frame boundaries: k_cascade_init* marks the start of a frame's stage loop; the prescale (tables + gather) sits just before it
inits = [i for i, n in enumerate(names) if n.startswith("k_cascade_init")]
if len(inits) < frames + 1:
    frames = max(1, len(inits) - 1)
lo, hi = inits[-frames - 1], inits[-1]
cur = rows[lo:hi]
per = collections.OrderedDict()
busy = gaps = 0
big = []
for i, r in enumerate(cur):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    k = short(r["Kernel_Name"])
    per.setdefault(k, [0, 0.0, []])
    per[k][0] += 1
    per[k][1] += d
    if len(per[k][2]) < 12:
        per[k][2].append(round(d / 1e3, 1))
    if i:
        g = int(r["Start_Timestamp"]) - int(cur[i - 1]["End_Timestamp"])
        if g > 0:
            gaps += g
            if g > 4000:
                big.append(g / 1e3)
span = (int(cur[-1]["End_Timestamp"]) - int(cur[0]["Start_Timestamp"])) / 1e3
print("%d frames: %.1f dispatches, span %.1f us, busy %.1f us, idle %.1f us per frame (gaps > 4 us: %.1f per frame, mean %.1f us)" % (
    frames, len(cur) / frames, span / frames, busy / 1e3 / frames, gaps / 1e3 / frames, len(big) / frames, sum(big) / max(1, len(big))))
print("launches per frame x total us per frame  kernel  [durations of the first launches, us]")
for k, (c, d, first) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%5.1f x %8.1f us  %-28s %s" % (c / frames, d / 1e3 / frames, k, first))
