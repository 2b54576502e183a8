"""TEST ORACLE — NOT PRODUCT CODE.  *** parity unpinned ***

CPU restatement (numpy, float64) of the execute semantics of the MDP / cuicuilco nodes on the
reference's hot path ``sl = networks[k].execute(subimages_arr, benchmark=benchmark)``
(FaceDetectUpdated.py:699; face_analysis.py:1064,1257).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product (``pyfaceanalysis_amd``) never does, and this module never
imports the product: it dispatches on class *names* so that it works on any object graph with
MDP-style attributes.

Why "parity unpinned": the arithmetic of this path is not in /root/reference at all.  It lives
in two un-vendored third-party packages — mdp-toolkit ("current master", no version pin,
README.md:19-20,30) and cuicuilco @ 9bfd24201b0e4107b9689c13b2da55e3a01cfb55 (README.md:20,31) —
neither present here, the trained flows (SavedNetworks/*.pckl) are stripped
(.MISSING_LARGE_BLOBS:1-8), and the reference has no tests or golden vectors for the path
(SURVEY.md §4, §8c).  What follows restates the published MDP behaviour of each node, anchored
on the reference's own call sites; each function cites the call site / evidence it follows.

The structure deliberately mirrors MDP's call granularity (a Python loop over the nodes of a
Layer, a fancy-index gather per Switchboard, one ``numpy.dot`` per linear node) because that is
what the reference's CPU cost consists of (SURVEY.md §3.2, §8d "CPU baseline beside it").
"""
from __future__ import annotations

import numpy as np


def _mro_names(obj):
    return [c.__name__ for c in type(obj).__mro__]


# --- leaf nodes -------------------------------------------------------------------------------

def exec_pca(node, x):
    """mdp.nodes.PCANode._execute / WhiteningNode: ``mult(x - avg, v)``.
    Alias evidence FaceDetectUpdated.py:63-65; SURVEY.md §8a row a5."""
    return np.dot(x - node.avg, node.v)


def exec_sfa(node, x):
    """mdp.nodes.SFANode._execute / cuicuilco GSFANode: ``mult(x, sf) - _bias``.
    Alias evidence FaceDetectUpdated.py:63,65; SURVEY.md §8a row a7."""
    return np.dot(x, node.sf) - node._bias


def exec_linreg(node, x):
    """mdp.nodes.LinearRegressionNode._execute (with_bias): ``[1, x] @ beta``."""
    return node.beta[0:1, :] + np.dot(x, node.beta[1:, :])


def apply_expfunc(f, x):
    """One cuicuilco.nonlinear_expansion function on a block."""
    d = x.shape[1]
    if f.sel > 0:
        x = x[:, :min(f.sel, d)]
        d = x.shape[1]
    if f.kind == "identity":
        return x
    if f.kind == "abs_pow":
        return np.abs(x) ** f.expo
    if f.kind == "signed_pow":
        return np.sign(x) * np.abs(x) ** f.expo
    if f.kind == "quadratic":
        cols = [x[:, i:i + 1] * x[:, i:] for i in range(d)]
        return np.concatenate(cols, axis=1) if cols else np.zeros((x.shape[0], 0))
    if f.kind == "pair_adj":
        k = f.k
        if d - k <= 0:
            return np.zeros((x.shape[0], 0))
        return x[:, :d - k] * x[:, k:]
    if f.kind == "pair_band":       # the other reading of pair_prodsadj{k}_ex: offsets 0 .. k-1, squares first
        cols = [x[:, :d - off] * x[:, off:] for off in range(f.k) if d - off > 0]
        return np.concatenate(cols, axis=1) if cols else np.zeros((x.shape[0], 0))
    raise ValueError("oracle: unknown expansion kind %r" % (f.kind,))


def exec_expansion(node, x):
    """cuicuilco.more_nodes.GeneralExpansionNode._execute: hstack of the functions
    (alias FaceDetectUpdated.py:62)."""
    return np.concatenate([apply_expfunc(f, x) for f in node.funcs], axis=1)


def exec_igsfa(node, x):
    """cuicuilco.igsfa_node.iGSFANode._execute (IEVMLRecNode in older pickles,
    face_analysis.py:463-467; alias FaceDetectUpdated.py:64); SURVEY.md §8a row a8.
    The two points on which public descriptions of cuicuilco differ are read from explicit fields of
    the node: ``scaling`` (per-column ``magn_n_sfa_x`` or a matrix) and ``lr_input`` (does the linear
    reconstruction read the scaled or the normalised slow features)."""
    x0 = x - node.x_mean
    e = execute_node(node.exp_node, x0) if node.exp_node is not None else x0
    n_sfa = execute_node(node.sfa_node, e)
    if getattr(node, "scaling", "per_column") == "matrix":
        s = np.dot(n_sfa, node.scaling_matrix)
    else:
        s = n_sfa * node.magn_n_sfa_x
    if node.reconstruct_with_sfa and node.lr_node is not None:
        lr_in = n_sfa if getattr(node, "lr_input", "scaled") == "unscaled" else s
        r = x0 - execute_node(node.lr_node, lr_in)
    else:
        r = x0
    q = execute_node(node.pca_node, r)
    return np.concatenate([s[:, :node.num_sfa_features_preserved], q], axis=1)


# --- structural nodes -------------------------------------------------------------------------

def exec_switchboard(node, x):
    """mdp.hinet.Switchboard._execute: ``x[:, connections]``."""
    return x[:, node.connections]


def exec_layer(node, x):
    """mdp.hinet.Layer._execute: per-node column slices, results written side by side
    (``Layer.nodes``: face_analysis.py:460-462)."""
    y = np.zeros((x.shape[0], node.output_dim), dtype=x.dtype)
    i0 = o0 = 0
    for sub in node.nodes:
        i1, o1 = i0 + sub.input_dim, o0 + sub.output_dim
        y[:, o0:o1] = execute_node(sub, x[:, i0:i1])
        i0, o0 = i1, o1
    return y


def exec_flownode(node, x):
    for sub in node.flow:
        x = execute_node(sub, x)
    return x


_DISPATCH = (
    ("iGSFANode", exec_igsfa),
    ("PCANode", exec_pca),             # also WhiteningNode (subclass)
    ("SFANode", exec_sfa),             # also GSFANode (subclass)
    ("LinearRegressionNode", exec_linreg),
    ("GeneralExpansionNode", exec_expansion),
    ("Switchboard", exec_switchboard),  # also PInvSwitchboard, Rectangular2dSwitchboard
    ("Layer", exec_layer),              # also CloneLayer
    ("FlowNode", exec_flownode),
    ("IdentityNode", lambda node, x: x),
    ("HeadNode", lambda node, x: x[:, :node.output_dim]),
    ("CutoffNode", lambda node, x: np.clip(x, node.lower_bound, node.upper_bound)),
)


def execute_node(node, x):
    """mdp.Node.execute: dimension check, then the class's ``_execute``."""
    if x.ndim != 2 or x.shape[1] != node.input_dim:
        raise ValueError("oracle: %s expects input_dim %d, got array of shape %r"
                         % (type(node).__name__, node.input_dim, x.shape))
    names = _mro_names(node)
    for cname, fn in _DISPATCH:
        if cname in names:
            return fn(node, x)
    raise TypeError("oracle: no restatement for node class %s" % type(node).__name__)


def execute_flow(flow_nodes, x, nodenr=None):
    """mdp.Flow.execute / _execute_seq: ``for node in flow[:nodenr+1]: x = node.execute(x)``
    (call site FaceDetectUpdated.py:699).  Input is cast to float64 as MDP nodes do."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError("oracle: execute expects a 2-d array")
    nodes = list(flow_nodes)
    if nodenr is not None:
        nodes = nodes[:nodenr + 1]
    for node in nodes:
        x = execute_node(node, x)
    return x
