"""Per launch position of a step: HBM read / write MB from two rocprofv3 --pmc passes (FETCH_SIZE x2, WRITE_SIZE; KiB -> bytes,
MI355X_MICROARCH.md §HBM).  python tools/pmc_traffic.py <label> <fetch dir> <write dir>"""
import collections, csv, glob, os, sys

label, dirs = sys.argv[1], sys.argv[2:4]


def short(n):
    for s in ("hg::fused::(anonymous namespace)::", "hg::(anonymous namespace)::", "hg::fused::", "(StageParams, StageParams)", "(StageParams)", "(TailParams)", "void "):
        n = n.replace(s, "")
    return n


def last_step(d, name):
    f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    seq = [(k, v) for (did, k), v in sorted(per.items()) if k.startswith("k_")]
    out, pos = collections.OrderedDict(), -1
    for k, v in seq:
        pos = 0 if k.startswith("k_stage0") else pos + 1
        out[(pos, k)] = v          # the last step's dispatch wins
    return out


rd, wr = last_step(dirs[0], "FETCH_SIZE"), last_step(dirs[1], "WRITE_SIZE")
print("== %s" % label)
tr = tw = 0.0
for (pos, k), v in rd.items():
    r, w = v * 1024 * 2 / 1e6, wr.get((pos, k), 0.0) * 1024 / 1e6
    tr += r
    tw += w
    print("  %d %-44s read %7.1f MB  write %7.1f MB" % (pos, k[:44], r, w))
print("  sum: read %.1f MB  write %.1f MB" % (tr, tw))
