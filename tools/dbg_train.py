import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyfaceanalysis_amd import synth
from oracle import mdp_restate
preset = sys.argv[1]
dev = synth.build_preset(preset, device=0)
host = synth.build_preset(preset)
x = synth.make_subimages(20, synth.preset_input_side(preset), seed=7, dtype=np.float64)
for k in range(1, len(host), 2):
    ya, yb = mdp_restate.execute_flow(host, x, nodenr=k), mdp_restate.execute_flow(dev, x, nodenr=k)
    d = np.minimum(np.abs(ya - yb), np.abs(ya + yb)).max(axis=0)
    n_nodes = len(host[k].nodes)
    s = ya.shape[1] // n_nodes
    per_col = d.reshape(n_nodes, s).max(axis=0) / np.abs(ya).max()
    print("layer %2d: nodes %4d, worst relative difference by output column:" % (k // 2, n_nodes), " ".join("%.0e" % v for v in per_col[:12]), "... max %.1e" % per_col.max())
