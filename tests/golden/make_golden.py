"""Generates the fixtures in this directory.  Run from the repo root:  python tests/golden/make_golden.py

1. flows_*.npz — inputs, flow blob and float64 outputs of small trained hierarchies, produced by
   the build's own oracle (oracle/mdp_restate.py).  The reference holds no golden vector for the
   hot path and its arithmetic cannot be run here, so these pin the ORACLE and
   the HIP path against regressions, not against the original program ("parity unpinned").
2. classifiers.npz — the PARAMETERS (data) of seven of the reference's own
   SavedClassifiers/*.pckl files (one per (classes, features) shape the 21 files come in: (2,5) (10,9) (39,4) (39,5)
   (50,10) (50,12) (50,20) — the K = 50, d = 20 pose regressors are the ones whose sqrt-determinants reach 1e42), read with a stub unpickler (no mdp import), plus oracle
   regression outputs on seeded inputs.  The stored `_sqrt_def_covs` of those files is the one
   reference-owned known answer near this path: it must equal det(inv_covs)^-1/2.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import mdp_restate, ref_c  # noqa: E402
from pyfaceanalysis_amd import synth  # noqa: E402
from pyfaceanalysis_amd.blob import flow_to_blob  # noqa: E402
from pyfaceanalysis_amd.classifier import load_stub_pickle  # noqa: E402


def flows():
    cases = {"t3l8": ("T3L-8", {}), "t5l16": ("T5L-16", {}), "t5l16_sep": ("T5L-16", {"layout": "separate"}),
             "t5l16_igsfa": ("T5L-16", {"node_kind": "igsfa"})}
    for key, (preset, kw) in cases.items():
        nodes = synth.build_preset(preset, **kw)
        side = synth.preset_input_side(preset)
        x = synth.make_subimages(24, side, seed=4242, dtype=np.uint8)
        y = mdp_restate.execute_flow(nodes, x)
        np.savez_compressed(os.path.join(HERE, "flow_%s.npz" % key), x=x, y=y,
                            blob=np.frombuffer(flow_to_blob(nodes), dtype=np.uint8))
        print(key, x.shape, y.shape)


def classifiers():
    files = sorted(glob.glob("/root/reference/SavedClassifiers/*.pckl"))
    picks = {}
    for f in files:
        o = load_stub_pickle(f)
        key = (np.asarray(o.means).shape)
        picks.setdefault(key, f)
    out = {}
    rng = np.random.default_rng(7)
    out["n_classifiers"] = np.array(len(picks))
    for i, (shape, f) in enumerate(sorted(picks.items())):
        o = load_stub_pickle(f)
        means, inv_covs = np.asarray(o.means, float), np.asarray(o.inv_covs, float)
        sd, p, avg = np.asarray(o._sqrt_def_covs, float), np.asarray(o.p, float), np.asarray(o.avg_labels, float)
        k, d = means.shape
        # inputs: class means + noise shaped by the class covariances (so posteriors are not all one-hot)
        cls = rng.integers(0, k, 40)
        x = np.stack([rng.multivariate_normal(means[c], np.linalg.inv(inv_covs[c]) * 4.0) for c in cls])
        # and rows far from every class (all densities underflow outside the log domain) / exactly on a class mean
        far = means[rng.integers(0, k, 4)] + rng.normal(size=(4, d)) * 30.0 * np.sqrt(np.abs(np.linalg.inv(inv_covs[0]).diagonal()))
        x = np.concatenate([x, far, means[:2]])
        reg, std = ref_c.gauss_regression(x, means, inv_covs, sd, p, avg)
        out.update({"c%d_means" % i: means, "c%d_inv_covs" % i: inv_covs, "c%d_sqrt_def_covs" % i: sd, "c%d_p" % i: p,
                    "c%d_avg_labels" % i: avg, "c%d_x" % i: x, "c%d_reg" % i: reg, "c%d_std" % i: std,
                    "c%d_source" % i: np.array(os.path.basename(f))})
        print("classifier", i, shape, os.path.basename(f)[:60])
    np.savez_compressed(os.path.join(HERE, "classifiers.npz"), **out)


if __name__ == "__main__":
    flows()
    if os.path.isdir("/root/reference/SavedClassifiers"):
        classifiers()
