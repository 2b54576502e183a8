"""Summarise a rocprofv3 kernel trace CSV: per kernel position in a call, duration and the idle gap before it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_call = int(sys.argv[2])
rows = rows[-per_call * 30:]           # the last 30 calls
dur = collections.defaultdict(list); gap = collections.defaultdict(list); names = {}
for i, r in enumerate(rows):
    k = i % per_call
    names[k] = r["Kernel_Name"][:60]
    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if i: gap[k].append(int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
tot_d = tot_g = 0
for k in range(per_call):
    d = sorted(dur[k])[len(dur[k]) // 2] / 1e3; g = sorted(gap[k])[len(gap[k]) // 2] / 1e3
    tot_d += d; tot_g += g
    print("%2d  %-60s  %7.1f us   gap before %6.1f us" % (k, names[k], d, g))
print("per call: kernels %.1f us + gaps %.1f us = %.1f us" % (tot_d, tot_g, tot_d + tot_g))
