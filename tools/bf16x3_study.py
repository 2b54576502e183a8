"""Numeric study for DESIGN.md §8(1): what would split-bf16 MFMA products cost in accuracy?

Runs the numpy oracle's forward pass with every matrix product replaced by an emulation of
  fp32  : operands rounded to fp32, fp32 accumulate      (what the kernels do today)
  bf16x6: three-way split, the six products down to 2^-16
  bf16x3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate
  bf16x2: a_hi*b_hi + a_lo*b_hi  (activation split only, weights bf16)
  bf16  : a_hi*b_hi
and reports max|d|/max|ref| against the float64 oracle.  CPU only; test infrastructure (imports oracle/).
"""
import os, sys, types
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import mdp_restate as M
from pyfaceanalysis_amd import synth, blob


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(x):
    x = x.astype(np.float32)
    hi = bf16(x)
    lo = bf16(x - hi)
    return hi, lo


MODE = "fp32"


def dot(a, b):
    a32 = np.asarray(a, dtype=np.float32)
    b32 = np.asarray(b, dtype=np.float32)
    if MODE == "f64":
        return np.dot(a, b)
    if MODE == "fp32":
        return np.dot(a32, b32).astype(np.float64)
    ah, al = split(a32)
    bh, bl = split(b32)
    if MODE == "bf16":
        return np.dot(ah, bh).astype(np.float64)
    if MODE == "bf16x2":
        return (np.dot(ah, bh) + np.dot(al, bh)).astype(np.float64)
    if MODE == "bf16x3":
        return (np.dot(ah, bh) + (np.dot(ah, bl) + np.dot(al, bh))).astype(np.float64)
    if MODE == "bf16x6":
        am = bf16(a32 - ah); al2 = bf16(a32 - ah - am)
        bm = bf16(b32 - bh); bl2 = bf16(b32 - bh - bm)
        return (np.dot(ah, bh) + (np.dot(ah, bm) + np.dot(am, bh)) +
                (np.dot(ah, bl2) + np.dot(al2, bh) + np.dot(am, bm))).astype(np.float64)
    raise ValueError(MODE)


class _NP(types.ModuleType):
    def __getattr__(self, k):
        return dot if k == "dot" else getattr(np, k)


def main():
    global MODE
    preset = sys.argv[1] if len(sys.argv) > 1 else "U11L-64"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    _, flow = synth.cached_preset_blob(preset)
    side = synth.preset_input_side(preset)
    x = synth.make_subimages(n, side, seed=5).astype(np.float64)
    ref = M.execute_flow(flow, x)
    M.np = _NP("npshim")
    try:
        for MODE in ("fp32", "bf16x6", "bf16x3", "bf16x2", "bf16"):
            y = M.execute_flow(flow, x)
            d = np.abs(y - ref)
            print("%-7s max|d|/max|ref| = %.3e   first 20 cols %.3e" %
                  (MODE, d.max() / np.abs(ref).max(), d[:, :20].max() / np.abs(ref[:, :20]).max()))
    finally:
        M.np = np


if __name__ == "__main__":
    main()
