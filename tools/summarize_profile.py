"""Turn the rocprofv3 outputs of tools/profile.sh (under gpurun_out/) into the committed summary
profiles/<tag>_summary.md + <tag>_kernel_stats.csv + <tag>_traffic.json.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE
from separate --pmc passes, in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads, so the read side is doubled."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(n):
    return n.replace("hg::(anonymous namespace)::", "").replace("hg::fused::", "").replace("(StageParams, StageParams)", "").replace("(StageParams)", "")


def is_first(k):
    return k.startswith("void k_stage0") or k.startswith("void k_stage01p")


def by_position(seq):
    """seq: [(kernel name, grid, payload)] in dispatch order -> OrderedDict[(pos, kernel, grid)] -> [payload]
    where pos = index of the dispatch inside its step (a step starts with the stage-0 kernel)."""
    out = collections.OrderedDict()
    pos = -1
    for k, grid, payload in seq:
        if not k.startswith("void k_"):
            continue
        pos = 0 if is_first(k) else pos + 1
        out.setdefault((pos, k, grid), []).append(payload)
    return out


def counters(d):
    f = max(glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest run
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Grid_Size"])
        per.setdefault(key, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
    seq = [(k, grid, c) for (did, k, grid), c in sorted(per.items(), key=lambda kv: kv[0][0])]
    return {key: v[-1] for key, v in by_position(seq).items()}     # last step's dispatch


stats = max(glob.glob(os.path.join(G, "prof_" + tag, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(P, tag + "_kernel_stats.csv"))
trace = max(glob.glob(os.path.join(G, "prof_" + tag, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = by_position([(short(r["Kernel_Name"]), r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                   for r in rows])
fetch, write = counters("pmc_fetch_" + tag), counters("pmc_write_" + tag)
sq, lds = counters("pmc_sq_" + tag), counters("pmc_lds_" + tag)
lines = ["# rocprofv3 summary %s — `bench.py --steps 200 --warmup 30` (4096 x 128x128 fp32, U11L-128, 1 x MI355X; counter passes: 20 steps after 10)" % tag, "",
         "Durations: `rocprofv3 --kernel-trace --stats` (average over all dispatches of the run).  HBM bytes: separate",
         "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, KiB -> bytes, FETCH_SIZE x2 (gfx950 reports half of wide",
         "coalesced reads, MI355X_MICROARCH.md §HBM).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),",
         "kernel cycles = GRBM_GUI_ACTIVE / 8.", "",
         "| # | kernel | grid | calls | avg us | HBM read MB | HBM write MB | GB/s | MFMA busy | LDS bank-conflict share |",
         "|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
for (pos, k, grid), v in dur.items():
    avg = sum(v) / len(v)
    fk = fetch.get((pos, k, grid), {}).get("FETCH_SIZE")
    wk = write.get((pos, k, grid), {}).get("WRITE_SIZE")
    rd = fk * 1024 * 2 if fk is not None else None
    wr = wk * 1024 if wk is not None else None
    s = sq.get((pos, k, grid), {})
    l = lds.get((pos, k, grid), {})
    busy = None
    if s.get("SQ_VALU_MFMA_BUSY_CYCLES") and l.get("GRBM_GUI_ACTIVE"):
        busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * l["GRBM_GUI_ACTIVE"] / 8.0)
    conf = l["SQ_LDS_BANK_CONFLICT"] / l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else None
    gbs = (rd + wr) / (avg * 1e-6) / 1e9 if rd is not None and wr is not None else None
    lines.append("| %d | `%s` | %s | %d | %.1f | %s | %s | %s | %s | %s |" % (
        pos, k.replace("void ", ""), grid, len(v), avg, "%.1f" % (rd / 1e6) if rd is not None else "-",
        "%.1f" % (wr / 1e6) if wr is not None else "-", "%.0f" % gbs if gbs else "-",
        "%.0f%%" % (100 * busy) if busy else "-", "%.0f%%" % (100 * conf) if conf is not None else "-"))
    if rd is not None and wr is not None:
        traffic["%d|%s" % (pos, k)] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "avg_us": avg}
lines += ["", "Total kernel time per step: %.1f us" % sum(sum(v) / len(v) for v in dur.values())]
open(os.path.join(P, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)
print("\n".join(lines))
