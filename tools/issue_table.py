"""profiles/<tag>_issue_table.md: per layer of U11L-128 what the algorithm needs and what the kernels ISSUE.

Planner side (host only, from hg_flow_describe): 16x16x4 and 4x4x1 MFMA instructions per 16-row tile, all nodes of the layer.
Counter side (profiles/<tag>_counters.json, written by tools/summarize_profile.py from the round's rocprofv3 --pmc passes): MFMA /
vector / transcendental wave instructions of every launch of a step.  Cycles per wave instruction on one SIMD: 32 for
v_mfma_f32_16x16x4_f32, 8 for v_mfma_f32_4x4x1_16B_f32 (SQ_VALU_MFMA_BUSY_CYCLES agrees: the front kernel's 24.8 cycles per MFMA
are 42 x 32 + 18 x 8 over 60 instructions), 2.3 for a full-rate vector instruction, 9 for a transcendental one
(tools/ubench/mfma_valu_overlap.hip).     python tools/issue_table.py r05"""
import json, os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import _Handle

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS, TILES = 4096, 256
LAYER_FLOPS = [1118208, 1351680, 1971200, 2918400, 1843200, 921600, 460800, 230400, 115200, 57600, 28800]
blob, nodes = synth.cached_preset_blob("U11L-128")
h = _Handle(blob)
desc = h.describe()
h.close()
stages = []
for line in desc.splitlines():
    m = re.search(r"fused stage (\d+).*?: (\d+) nodes, K-blocks (\d+), tiles (\d+)x(\d+), (\d+) MFMA/tile \(issued: (\d+) x 16x16x4 \+ (\d+) x 4x4x1\)", line)
    if m:
        stages.append(dict(zip(("stage", "nodes", "kb1", "mt1", "mt2", "tiles16", "m16", "m4"), map(int, m.groups()))))
assert len(stages) == 11, desc
ctr = json.load(open(os.path.join(ROOT, "profiles", tag + "_counters.json")))
launch_of = {}
for key, v in ctr.items():
    for l in v["layers"]:
        launch_of[l] = (key, v)
out = ["# Issued against algorithmic work per layer — U11L-128, 4096 rows (%s)" % tag, "",
       "Planner columns are exact (host side, `hg_flow_describe`); counter columns come from the round's `rocprofv3 --pmc` passes",
       "(`profiles/%s_counters.json`; SQ counters are summed over all waves of a launch).  A *node visit* = one wave taking one node through" % tag,
       "its T batch tiles (front kernel: one layer-1 node with its two layer-0 children, T = 1; k_stage: T = 2; last launches: see kernel).", "",
       "| layer | nodes | algorithmic FLOP / row | MFMA16 / tile | MFMA4 / tile | issued FLOP / row | issued / algorithmic | as 16x16 only (round 4's count) | MFMA cycles / node / tile |",
       "|---|---|---|---|---|---|---|---|---|"]
tot_alg = tot_iss = tot_r4 = 0
for l, st in enumerate(stages):
    iss = (st["m16"] * 2048 + st["m4"] * 512) / 16.0
    r4 = st["tiles16"] * 2048 / 16.0
    tot_alg += LAYER_FLOPS[l]; tot_iss += iss; tot_r4 += r4
    cyc = (st["m16"] * 32 + st["m4"] * 8) / st["nodes"]
    out.append("| %d | %d | %d | %d | %d | %d | %.3f | %d | %.0f |" % (l, st["nodes"], LAYER_FLOPS[l], st["m16"], st["m4"], iss, iss / LAYER_FLOPS[l], r4, cyc))
out.append("| all | | %d | | | %d | %.3f | %d | |" % (tot_alg, tot_iss, tot_iss / tot_alg, tot_r4))
out += ["", "Where the padding sits: 16-row tiles (13 -> 16 rows in both affines of layer 0: 1.41x; 60 -> 64 rows in layers 3-10: every fourth m-tile holds 12",
        "rows) and 4-deep k-steps (35 = 32 + 3 inputs of a layer-3 child: its last k-step is one value).  Tiles of <= 4 real rows (layers 1-2) run on",
        "4x4x1: 512 instead of 2048 issued FLOP each.  Round 4's `padded_flops_per_subimage` (13 731 840) counted those tiles as 16x16 too; issued work is",
        "the column above, %d FLOP per row = %.3f x algorithmic." % (tot_iss, tot_iss / tot_alg), "",
        "## Per launch: instructions the counters saw, and the issue bound", "",
        "| launch | layers | us | MFMA instr / visit (planner: 16x16 + 4x4) | MFMA cycles / visit | vector instr / visit | transcendental / visit | vector + transcendental cycles / visit | issue bound MFMA / (MFMA + vector) | achieved of 157.3 TFLOP/s |",
        "|---|---|---|---|---|---|---|---|---|---|"]
for key, v in ctr.items():
    ls = v["layers"]
    if not ls or v.get("SQ_INSTS_MFMA") is None:
        continue
    k = key.split("|", 1)[1]
    if "k_stage01d" in k:
        visits = stages[1]["nodes"] * TILES
    elif "k_tail" in k:
        visits = TILES * 16          # one wave = (node, m-tile) of the widest fused layer, one tile
    else:
        T = 1 if re.search(r"k_stage<\d+, \d+, 1,", k) else 2
        visits = stages[ls[0]]["nodes"] * TILES / T
    n_mfma, n_tr = v["SQ_INSTS_MFMA"], v["SQ_INSTS_VALU_TRANS_F32"]
    n_valu = v["SQ_INSTS_VALU"] - n_mfma - n_tr
    m16 = sum(stages[l]["m16"] for l in ls) * TILES
    m4 = sum(stages[l]["m4"] for l in ls) * TILES
    mcyc = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    vcyc = 2.3 * n_valu + 9.0 * n_tr
    ach = sum(LAYER_FLOPS[l] for l in ls) * ROWS / (v["avg_us"] * 1e-6) / 157.3e12
    out.append("| `%s` | %s | %.1f | %.1f (%.1f + %.1f) | %.0f | %.1f | %.1f | %.0f | %.0f%% | %.0f%% |" % (
        k.replace("void ", "")[:44], "-".join(map(str, (ls[0], ls[-1]))) if len(ls) > 1 else ls[0], v["avg_us"], n_mfma / visits, m16 / visits, m4 / visits,
        mcyc / visits, n_valu / visits, n_tr / visits, vcyc / visits, 100 * mcyc / (mcyc + vcyc), 100 * ach))
out += ["", "(The planner's instruction counts and `SQ_INSTS_MFMA` agree launch by launch; `SQ_VALU_MFMA_BUSY_CYCLES` = 32 x MFMA16 + 8 x MFMA4.)"]
open(os.path.join(ROOT, "profiles", tag + "_issue_table.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
