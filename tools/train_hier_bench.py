"""Whole-hierarchy training on the GPU (synth.train_hierarchy(device=0)) against the numpy trainer: wall time per preset."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth

for preset in sys.argv[1:] or ["T5L-16", "U11L-64", "U11L-128"]:
    synth.build_preset("T3L-8", device=0)          # warm-up (library, rocSOLVER handles)
    t0 = time.perf_counter(); dev = synth.build_preset(preset, device=0); td = time.perf_counter() - t0
    t0 = time.perf_counter(); host = synth.build_preset(preset); th = time.perf_counter() - t0
    from oracle import mdp_restate
    x = synth.make_subimages(20, synth.preset_input_side(preset), seed=7, dtype=np.float64)
    ya, yb = mdp_restate.execute_flow(host, x), mdp_restate.execute_flow(dev, x)
    err = float(np.minimum(np.abs(ya - yb).max(axis=0), np.abs(ya + yb).max(axis=0)).max() / np.abs(ya).max())
    print("%s: trained on the GPU in %.2f s, numpy (host cores) %.2f s; outputs of the two nets differ by %.1e" % (preset, td, th, err), flush=True)
