"""Summarise a rocprofv3 kernel trace of tools/frame_trace.py: per-kernel-name totals and idle gaps of the last frames."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames start with the prescale kernel: find its name = kernel of the first dispatch in a repeating pattern
names = [r["Kernel_Name"] for r in rows]
first = None
for i, r in enumerate(rows):
    if "prescale" in r["Kernel_Name"] or "k_resize" in r["Kernel_Name"] or "k_extent" in r["Kernel_Name"]:
        first = r["Kernel_Name"]; break
starts = [i for i, nm in enumerate(names) if nm == first]
# use dispatch pattern: split at large repetition of the first kernel name followed by many kernels
frames = []
cur = []
for i, r in enumerate(rows[starts[len(starts) // 2]:]):
    cur.append(r)
per = collections.defaultdict(lambda: [0, 0.0])
t_first, t_last, busy, gaps, big = int(cur[0]["Start_Timestamp"]), int(cur[-1]["End_Timestamp"]), 0, 0, []
for i, r in enumerate(cur):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    k = r["Kernel_Name"].split("(")[0][-60:]
    per[k][0] += 1; per[k][1] += d
    if i:
        g = int(r["Start_Timestamp"]) - int(cur[i - 1]["End_Timestamp"])
        if g > 0:
            gaps += g
            if g > 5000: big.append(g / 1e3)
print("dispatches %d over %.1f us: busy %.1f us, idle %.1f us (gaps > 5 us: %d, mean %.1f us)" % (len(cur), (t_last - t_first) / 1e3, busy / 1e3, gaps / 1e3, len(big), sum(big) / max(1, len(big))))
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%6d x %8.1f us total  %s" % (c, d / 1e3, k))
