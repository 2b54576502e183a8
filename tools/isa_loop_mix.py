"""Instruction mix of the front kernel's pass loop from the compiler's ISA listing (where do the vector
instructions between the MFMAs come from?).

    python tools/isa_loop_mix.py [--keep DIR]

compiles pyfaceanalysis_amd/csrc/hg_fused_front.hip with --save-temps, takes k_stage01d<float, false, true> (the instantiation
the headline runs), cuts out the pass loop (the outermost loop of the kernel) and prints per class — MFMA, transcendental,
other vector ALU, LDS, vector memory, scalar — the count and an attribution of the vector instructions to their source
constructs (by opcode and by the value they produce).  Counts are per pass and wave over ALL paths of the loop (grabber and
follower branches of the tile queue, error paths), so the scalar / readfirstlane figures are upper bounds for one wave."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pyfaceanalysis_amd", "csrc", "hg_fused_front.hip")
KERNEL = "_ZN2hg5fused10k_stage01dIfLb0ELb1EEEvNS0_11StageParamsES2_"


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    d = keep or tempfile.mkdtemp()
    os.makedirs(d, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-c", SRC, "-o", os.path.join(d, "front.o"),
                           "--save-temps"], cwd=d, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "hg_fused_front-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    k = lines[a:b]
    hdr = next(i for i, l in enumerate(k) if "Loop Header: Depth=1" in l)
    name = re.match(r"\.(LBB\d+_\d+)", k[hdr]).group(1)
    tag = name[1:] if name.startswith("L") else name
    last = max(i for i, l in enumerate(k) if ("Header=" + tag[1:] in l) or ("Parent Loop " + tag[1:] in l) or ("Header=" + tag in l) or ("Parent Loop " + tag in l))
    while last + 1 < len(k) and not k[last + 1].startswith(".LBB"):
        last += 1
    body = [l.strip() for l in k[hdr:last + 1]]
    body = [l for l in body if l and not l.startswith(";") and not l.startswith(".")]
    cls = collections.Counter()
    ops = collections.Counter()
    for l in body:
        op = l.split()[0]
        if op.startswith("v_mfma"):
            c = "MFMA 16x16x4" if "16x16x4" in op else "MFMA 4x4x1"
        elif op in ("v_exp_f32_e32", "v_log_f32_e32", "v_exp_f32_e64", "v_log_f32_e64"):
            c = "transcendental"
        elif op.startswith("v_"):
            c = "vector ALU"
        elif op.startswith("ds_"):
            c = "LDS"
        elif op.startswith(("buffer_", "global_", "flat_")):
            c = "vector memory"
        elif op.startswith("s_waitcnt"):
            c = "s_waitcnt"
        elif op.startswith("s_nop"):
            c = "s_nop"
        else:
            c = "scalar"
        cls[c] += 1
        if c in ("vector ALU", "transcendental"):
            ops[op] += 1
    print("pass loop of k_stage01d<float, WGQ> (%s .. ): %d instructions" % (name, len(body)))
    for c, n in cls.most_common():
        print("  %-16s %4d" % (c, n))
    print("vector instructions by opcode:")
    for op, n in ops.most_common():
        print("  %4d  %s" % (n, op))
    g = lambda *names: sum(ops[n] for n in names)
    rows = [
        ("|x|^p of the expansion: v_log + v_mul + v_exp per value (13 values: 2 x 4 layer 0, 4 + 1 layer 1)", g("v_log_f32_e32", "v_log_f32_e64") + g("v_exp_f32_e32", "v_exp_f32_e64") + g("v_mul_f32_e32", "v_mul_f32_e64")),
        ("mean subtraction of the input (2 children x 4 registers)", g("v_sub_f32_e32", "v_sub_f32_e64")),
        ("remainder tiles: lane-group reduce-scatter (permlane swaps)", g("v_permlane16_swap_b32_e32", "v_permlane32_swap_b32_e32")),
        ("remainder tiles and their accumulation: v_add_f32", g("v_add_f32_e32", "v_add_f32_e64")),
        ("register moves (zeroed 4x4 accumulators, results parked for the deferred store, 64-bit pairs)", g("v_mov_b32_e32", "v_mov_b64_e32", "v_accvgpr_write_b32", "v_accvgpr_read_b32")),
        ("wave-uniform values to scalars (tile queue, ring entries)", g("v_readfirstlane_b32")),
        ("compares / selects (lane 0 predicates, row bounds)", sum(n for o, n in ops.items() if o.startswith(("v_cmp", "v_cndmask")))),
    ]
    acc = 0
    print("attribution of the vector instructions:")
    for what, n in rows:
        print("  %4d  %s" % (n, what))
        acc += n
    print("  %4d  other" % (sum(ops.values()) - acc))


if __name__ == "__main__":
    main()
