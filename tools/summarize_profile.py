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
    return n.replace("hg::fused::(anonymous namespace)::", "").replace("(TailParams)", "").replace("hg::(anonymous namespace)::", "").replace("hg::fused::", "").replace("(StageParams, StageParams)", "").replace("(StageParams)", "")


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
    fs = glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv"))
    if not fs:
        return {}
    f = max(fs, key=os.path.getmtime)   # newest run
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Grid_Size"])
        c = per.setdefault(key, collections.Counter())
        c[r["Counter_Name"]] += float(r["Counter_Value"])
        c["_dispatch_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # of THIS dispatch, under counter collection
    seq = [(k, grid, c) for (did, k, grid), c in sorted(per.items(), key=lambda kv: kv[0][0])]
    return {key: v[-1] for key, v in by_position(seq).items()}     # last step's dispatch


stats = max(glob.glob(os.path.join(G, "prof_" + tag, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(P, tag + "_kernel_stats.csv"))
trace = max(glob.glob(os.path.join(G, "prof_" + tag, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = by_position([(short(r["Kernel_Name"]), r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                   for r in rows])
fetch, write = counters("pmc_fetch_" + tag), counters("pmc_write_" + tag)
sq, lds, mix = counters("pmc_sq_" + tag), counters("pmc_lds_" + tag), counters("pmc_mix_" + tag)
# algorithmic FLOPs per sub-image of every layer of U11L-128 and the layers each launch position covers
LAYER_FLOPS = [1118208, 1351680, 1971200, 2918400, 1843200, 921600, 460800, 230400, 115200, 57600, 28800]
ROWS, PEAK = 4096, 157.3e12
N_SE, N_SIMD = 32, 1024          # 8 XCDs x 4 shader engines (SQ counters are summed over them); 256 CUs x 4 SIMDs
# cycles per wave instruction on the vector ALU, measured (tools/ubench/mfma_valu_overlap.hip): full-rate 2.3, transcendental 9
CYC_VALU, CYC_TRANS = 2.3, 9.0
positions = list(dur.keys())
layer_of, nxt = {}, 0
for i, (pos, k, grid) in enumerate(positions):
    if "k_unpack" in k:
        layer_of[pos] = []
        continue
    n_l = 2 if is_first(k) and "01" in k else (len(LAYER_FLOPS) - nxt if "k_tail" in k else 1)
    layer_of[pos] = list(range(nxt, min(nxt + n_l, len(LAYER_FLOPS))))
    nxt += n_l
lines = ["# rocprofv3 summary %s — `bench.py --steps 200 --warmup 30` (4096 x 128x128 fp32, U11L-128, 1 x MI355X; counter passes: 20 steps after 10)" % tag, "",
         "Durations: `rocprofv3 --kernel-trace --stats` (average over all dispatches of the run).  HBM bytes: separate",
         "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, KiB -> bytes, FETCH_SIZE x2 (gfx950 reports half of wide",
         "coalesced reads, MI355X_MICROARCH.md §HBM).", "",
         "**Cycle base of `MFMA busy`.**  `SQ_VALU_MFMA_BUSY_CYCLES` counts 32 cycles per `v_mfma_f32_16x16x4_f32` (it equals",
         "8 x `SQ_INSTS_VALU_MFMA_MOPS_F32`, which counts 512 FLOPs per unit), summed over the 1024 SIMDs.  The kernel's own cycles are",
         "`SQ_BUSY_CYCLES / 32` (the counter is summed over the 32 shader engines; cycles during which the SQ holds waves); divided by the",
         "dispatch's duration in the same pass this gives the clock the chip held (column `GHz`).  Round 2 divided by `GRBM_GUI_ACTIVE / 8`",
         "instead, which also covers the dispatch's set-up and drain under counter collection and reads 10-25 % longer than the kernel",
         "(column `busy (GRBM base)`, for comparison) — that, not the counter, was the gap between 55 % and the FLOP-derived 64 % for the",
         "front kernel.", "",
         "**Issue bound.**  fp32 MFMAs and vector-ALU instructions share one issue path on gfx950 (measured: they serialise one for one,",
         "`tools/ubench/mfma_valu_overlap.hip`; `SQ_VALU_MFMA_COEXEC_CYCLES` below is the counter's own view of the overlap), so a kernel cannot",
         "spend more than MFMA cycles / (MFMA + VALU cycles) of its time in the matrix pipe.  VALU cycles = 2.3 x (non-MFMA, non-transcendental",
         "vector instructions) + 9 x (transcendental ones), per wave instruction, from `SQ_INSTS_VALU`, `SQ_INSTS_MFMA`, `SQ_INSTS_VALU_TRANS_F32`.",
         "`bound x clock` = issue bound x GHz / 2.4 x (algorithmic / issued FLOPs): the share of the 157.3 TFLOP/s spec peak (2.4 GHz, no",
         "padding) a kernel could reach if it did nothing but issue; `achieved` = algorithmic FLOPs / duration / 157.3 TFLOP/s.", "",
         "| # | kernel | grid | calls | avg us | HBM read MB | HBM write MB | GB/s | GHz | MFMA busy | busy (GRBM base) | VALU : TRANS : MFMA instr (M) | co-exec | issue bound | bound x clock | achieved | LDS bank-conflict share |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
counters_out = {}      # per launch position: the raw counters the issue table is built from (tools/issue_table.py)
for (pos, k, grid), v in dur.items():
    avg = sum(v) / len(v)
    fk = fetch.get((pos, k, grid), {}).get("FETCH_SIZE")
    wk = write.get((pos, k, grid), {}).get("WRITE_SIZE")
    rd = fk * 1024 * 2 if fk is not None else None
    wr = wk * 1024 if wk is not None else None
    s = sq.get((pos, k, grid), {})
    l = lds.get((pos, k, grid), {})
    m = mix.get((pos, k, grid), {})
    busy = busy_grbm = ghz = None
    if s.get("SQ_VALU_MFMA_BUSY_CYCLES") and s.get("SQ_BUSY_CYCLES"):
        cyc = s["SQ_BUSY_CYCLES"] / N_SE
        busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
        ghz = cyc / (s["_dispatch_us"] * 1e3)
    if s.get("SQ_VALU_MFMA_BUSY_CYCLES") and l.get("GRBM_GUI_ACTIVE"):
        busy_grbm = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * l["GRBM_GUI_ACTIVE"] / 8.0)
    flops = sum(LAYER_FLOPS[i] for i in layer_of.get(pos, [])) * ROWS
    ach = flops / (avg * 1e-6) / PEAK if flops else None
    mixs = coex = ib = ibc = None
    if m.get("SQ_INSTS_VALU") and s.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        n_mfma, n_tr = m.get("SQ_INSTS_MFMA", 0.0), m.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        n_valu = m["SQ_INSTS_VALU"] - n_mfma - n_tr
        mixs = "%.1f : %.1f : %.1f" % (n_valu / 1e6, n_tr / 1e6, n_mfma / 1e6)
        mf = s["SQ_VALU_MFMA_BUSY_CYCLES"]
        ib = mf / (mf + CYC_VALU * n_valu + CYC_TRANS * n_tr)
        if m.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
            coex = m["SQ_VALU_MFMA_COEXEC_CYCLES"] / mf
        issued = s.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
        if ghz and issued and flops:
            ibc = ib * ghz / 2.4 * flops / issued
    conf = l["SQ_LDS_BANK_CONFLICT"] / l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else None
    gbs = (rd + wr) / (avg * 1e-6) / 1e9 if rd is not None and wr is not None else None
    pct = lambda x: "%.0f%%" % (100 * x) if x is not None else "-"
    lines.append("| %d | `%s` | %s | %d | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        pos, k.replace("void ", ""), grid, len(v), avg, "%.1f" % (rd / 1e6) if rd is not None else "-",
        "%.1f" % (wr / 1e6) if wr is not None else "-", "%.0f" % gbs if gbs else "-", "%.2f" % ghz if ghz else "-",
        pct(busy), pct(busy_grbm), mixs or "-", pct(coex), pct(ib), pct(ibc), pct(ach), pct(conf)))
    if rd is not None and wr is not None:
        traffic["%d|%s" % (pos, k)] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "avg_us": avg}
    counters_out["%d|%s" % (pos, k)] = {"avg_us": avg, "grid": grid, "layers": layer_of.get(pos, []), "ghz": ghz,
                                        **{n: m.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32")},
                                        **{n: s.get(n) for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_BUSY_CYCLES", "SQ_WAVES")}}
tot = sum(sum(v) / len(v) for v in dur.values())
top = sum(sum(v) / len(v) for (pos, k, grid), v in dur.items() if pos >= 5)
lines += ["", "Total kernel time per step: %.1f us; layers 6-10 and the row-major output (positions >= 5): %.1f us" % (tot, top)]
open(os.path.join(P, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)
json.dump(counters_out, open(os.path.join(P, tag + "_counters.json"), "w"), indent=1)
print("\n".join(lines))
