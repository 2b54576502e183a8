"""Offline converter: pickled MDP / cuicuilco flow  ->  description objects  ->  neutral blob,
WITHOUT importing mdp or cuicuilco.

The reference loads its networks with ``cache_obj.load_obj_from_cache`` = ``pickle.load`` of an
``mdp.Flow`` object graph (face_analysis.py:457, 473-478), which needs mdp-toolkit, cuicuilco and
the module aliases of FaceDetectUpdated.py:57-68 on the path.  Here the pickle is read with
``classifier.StubUnpickler``: every non-numpy global becomes an attribute bag that remembers its
module and class name, and this module maps those bags, by class NAME and MDP attribute names,
onto ``pyfaceanalysis_amd.nodes``.  Unknown node classes or expansion functions raise, and so does
everything about a node that this converter does not understand and that could change its execute:
an iGSFANode with a ``slow_feature_scaling_method`` outside the whitelist below, with numeric state the
conversion does not consume, or whose linear-reconstruction input (scaled / unscaled slow features —
public descriptions of cuicuilco differ, see ``nodes.iGSFANode``) the caller has not stated; a
LinearRegressionNode without an intercept row.  The trained flows themselves are not shipped with the reference
(.MISSING_LARGE_BLOBS), so this is exercised on synthetic pickles built under the same module /
class names (tests/test_pickle_import.py).
"""
from __future__ import annotations

import re

import numpy as np

from . import nodes as N
from .blob import flow_to_blob
from .classifier import load_stub_pickle

# cuicuilco.nonlinear_expansion function names -> ExpFunc  
_FUNCS = {
    "identity": N.identity, "I": N.identity,
    "unsigned_08expo": N.unsigned_08expo, "signed_08expo": N.signed_08expo,
    "QT": N.QT,
    "unsigned_2expo": N.unsigned_expo(2.0), "signed_2expo": N.signed_expo(2.0),
    "unsigned_06expo": N.unsigned_expo(0.6), "signed_06expo": N.signed_expo(0.6),
    "unsigned_09expo": N.unsigned_expo(0.9), "signed_09expo": N.signed_expo(0.9),
}


def _cls(obj):
    return obj.__name__ if isinstance(obj, type) else type(obj).__name__


def _get(obj, *names, **kw):
    d = obj.__dict__
    for n in names:
        if n in d and d[n] is not None:
            return d[n]
    if "default" in kw:
        return kw["default"]
    raise ValueError("pickled %s has none of the attributes %s (has: %s)" % (_cls(obj), names, sorted(d)))


def _dims(obj):
    return int(_get(obj, "_input_dim", "input_dim")), int(_get(obj, "_output_dim", "output_dim"))


_PAIR_NAME = re.compile(r"^pair_prodsadj(\d+)_ex$")


def convert_func(f, pair_prodsadj_reading=None):
    """One pickled expansion function (a global of cuicuilco.nonlinear_expansion, identified by NAME).
    ``pair_prodsadj{k}_ex`` has two possible meanings (``nodes.pair_prodsadj_ex``); the caller must state which."""
    name = getattr(f, "__name__", None) or _cls(f)
    m = _PAIR_NAME.match(name)
    if m:
        if pair_prodsadj_reading not in N.PAIR_READINGS:
            raise ValueError("expansion function %r: state pair_prodsadj_reading='offset' (x_i * x_{i+k} only) or 'band' (offsets "
                             "0 .. k-1, squares included) — the rule lives in cuicuilco @9bfd242, which is not available here, "
                             "and the two give different (and differently wide) expansions" % name)
        return N.pair_prodsadj_ex(int(m.group(1)), pair_prodsadj_reading)
    if name not in _FUNCS:
        raise TypeError("unsupported expansion function %r (module %r)" % (name, getattr(f, "__module__", "?")))
    return _FUNCS[name]


# iGSFANode.slow_feature_scaling_method -> how the scaled slow features are formed from the normalised ones.
# [K] restated from public knowledge of cuicuilco's igsfa_node.py; anything else is refused.
_IGSFA_SCALING = {
    None: "none",                       # s = n
    "sensitivity_based": "per_column",  # s = n * magn_n_sfa_x
    "data_dependent": "per_column",
    "QR_decomposition": "matrix",       # s = n @ R.T
}
# attributes an iGSFANode stub may carry that do not enter execute (training configuration / bookkeeping)
_IGSFA_IGNORABLE = frozenset((
    "_input_dim", "_output_dim", "_dtype", "input_dim", "output_dim", "dtype", "_train_phase", "_train_phase_started",
    "_training", "_train_seq", "verbose", "pre_expansion_node_class", "pre_expansion_out_dim", "expansion_funcs",
    "expansion_output_dim", "expansion_starting_point", "max_length_slow_part", "max_num_samples_for_ev",
    "max_test_samples_for_ev", "offsetting_mode", "max_preserved_sfa", "out_sfa_filter", "delta_threshold",
    "evar", "first_call", "sfa_x_mean", "sfa_x_std", "expanded_dim", "num_sfa_features_preserved_max"))


def _is_numeric_state(v):
    if isinstance(v, bool) or v is None:
        return False
    if isinstance(v, (int, float, complex, np.number)):
        return True
    return isinstance(v, np.ndarray) and v.dtype.kind in "fiuc"


MATRIX_ORIENTATIONS = ("n@R.T", "n@R")


def convert_node(obj, igsfa_lr_input=None, ignore_attrs=(), pair_prodsadj_reading=None, igsfa_matrix_orientation=None):
    """One pickled node (stub) -> a pyfaceanalysis_amd.nodes object.

    ``igsfa_lr_input``: "scaled" or "unscaled" — which slow features a pickled iGSFANode's ``lr_node`` reads
    (``nodes.iGSFANode``).  Required as soon as an iGSFANode has both a reconstruction and a non-trivial
    scaling; there is no default because the reference's source for it is not available.
    ``ignore_attrs``: names of extra numeric attributes of iGSFANode stubs to accept unconsumed.
    ``pair_prodsadj_reading``: "offset" or "band" — the meaning of ``pair_prodsadj{k}_ex`` (``nodes.pair_prodsadj_ex``);
    required as soon as a pickled expansion names such a function, no default for the same reason.
    ``igsfa_matrix_orientation``: "n@R.T" or "n@R" — how the QR-scaling matrix ``R`` of an iGSFANode with
    ``slow_feature_scaling_method="QR_decomposition"`` meets the normalised slow features n (s = n @ R.T or s = n @ R); R is
    square, so the dimensions cannot tell, and the rule lives in cuicuilco @9bfd242: required for such nodes, no default."""
    if obj is None:
        return None
    kw = dict(igsfa_lr_input=igsfa_lr_input, ignore_attrs=ignore_attrs, pair_prodsadj_reading=pair_prodsadj_reading,
              igsfa_matrix_orientation=igsfa_matrix_orientation)
    name = _cls(obj)
    if name in ("PCANode", "WhiteningNode"):
        i, o = _dims(obj)
        v = np.asarray(_get(obj, "v"), dtype=np.float64)[:, :o]
        cls = N.WhiteningNode if name == "WhiteningNode" else N.PCANode
        return cls(np.asarray(_get(obj, "avg"), dtype=np.float64).reshape(-1), v)
    if name in ("SFANode", "GSFANode", "SFAPCANode"):
        i, o = _dims(obj)
        sf = np.asarray(_get(obj, "sf"), dtype=np.float64)[:, :o]
        avg = np.asarray(_get(obj, "avg"), dtype=np.float64).reshape(-1)
        bias = _get(obj, "_bias", default=None)
        bias = None if bias is None else np.asarray(bias, dtype=np.float64).reshape(-1)[:o]
        cls = N.GSFANode if name == "GSFANode" else N.SFANode
        return cls(avg, sf, bias)
    if name == "LinearRegressionNode":
        i, o = _dims(obj)
        beta = np.asarray(_get(obj, "beta"), dtype=np.float64)
        with_bias = _get(obj, "with_bias", default=True)
        if not with_bias or beta.shape != (i + 1, o):
            raise ValueError("LinearRegressionNode: only with_bias=True is covered (beta row 0 = intercept); the pickle has "
                             "with_bias=%r and beta of shape %r for %d -> %d" % (with_bias, beta.shape, i, o))
        return N.LinearRegressionNode(beta)
    if name == "GeneralExpansionNode":
        i, o = _dims(obj)
        node = N.GeneralExpansionNode([convert_func(f, pair_prodsadj_reading) for f in _get(obj, "funcs")], i)
        if node.output_dim != o:
            raise ValueError("GeneralExpansionNode: converted width %d != pickled output_dim %d%s" % (
                node.output_dim, o, " (pair_prodsadj_reading=%r gives the wrong width: try the other reading)" % pair_prodsadj_reading
                if any(f.kind in ("pair_adj", "pair_band") for f in node.funcs) else ""))
        return node
    if name in ("iGSFANode", "IEVMLRecNode"):
        consumed = {"x_mean", "exp_node", "sfa_node", "pca_node", "lr_node", "magn_n_sfa_x", "num_sfa_features_preserved",
                    "reconstruct_with_sfa", "slow_feature_scaling_method", "R"}
        exp = convert_node(_get(obj, "exp_node", default=None), **kw)
        sfa = convert_node(_get(obj, "sfa_node"), **kw)
        pca = convert_node(_get(obj, "pca_node"), **kw)
        rec = bool(_get(obj, "reconstruct_with_sfa", default=True))
        lr = convert_node(_get(obj, "lr_node", default=None), **kw) if rec else None
        k = int(_get(obj, "num_sfa_features_preserved", default=sfa.output_dim))
        method = obj.__dict__.get("slow_feature_scaling_method", "sensitivity_based" if "magn_n_sfa_x" in obj.__dict__ else None)
        if method not in _IGSFA_SCALING:
            raise ValueError("iGSFANode: slow_feature_scaling_method=%r is not covered (known: %s); refusing to convert rather "
                             "than guess its execute" % (method, sorted(m for m in _IGSFA_SCALING if m)))
        scaling, magn, matrix = _IGSFA_SCALING[method], None, None
        if scaling == "matrix":
            if igsfa_matrix_orientation not in MATRIX_ORIENTATIONS:
                raise ValueError("iGSFANode (%d -> %d) with slow_feature_scaling_method='QR_decomposition': state "
                                 "igsfa_matrix_orientation='n@R.T' or 'n@R' (CLI: --igsfa-matrix-orientation) — how the pickled square "
                                 "matrix R meets the normalised slow features.  The rule lives in cuicuilco @9bfd242, which is not "
                                 "available here, and the two give different features.  (Until round 3 such nodes were imported "
                                 "silently as n@R.T: callers of convert_node / load_flow_pickle / pickle_to_blob that relied on that "
                                 "pass igsfa_matrix_orientation='n@R.T' to keep the old result.)" % _dims(obj))
            matrix = np.asarray(_get(obj, "R"), dtype=np.float64)
            if matrix.ndim != 2 or matrix.shape[0] != matrix.shape[1]:
                raise ValueError("iGSFANode: QR scaling matrix R of shape %r is not square" % (matrix.shape,))
            if igsfa_matrix_orientation == "n@R.T":
                matrix = matrix.T                                              # nodes.iGSFANode computes s = n @ scaling_matrix
        elif scaling == "per_column":
            magn = np.asarray(_get(obj, "magn_n_sfa_x"), dtype=np.float64).reshape(-1)
        else:
            scaling, magn = "per_column", np.ones(sfa.output_dim)
        stray = sorted(a for a, v in obj.__dict__.items()
                       if a not in consumed and a not in _IGSFA_IGNORABLE and a not in ignore_attrs and _is_numeric_state(v))
        if stray:
            raise ValueError("iGSFANode: pickled numeric state %s is not consumed by the conversion (assumed execute rule: "
                             "s = sfa(exp(x - x_mean)) scaled by %s; r = x0 - lr(s or n); y = [s[:k], pca(r)]); pass "
                             "ignore_attrs=%r if it does not enter execute" % (stray, method, tuple(stray)))
        trivial = matrix is None and bool(np.all(magn == 1.0))
        if lr is not None and rec and not trivial and igsfa_lr_input not in N.iGSFANode.LR_INPUTS:
            raise ValueError("iGSFANode with a linear reconstruction and %s scaling: state igsfa_lr_input='scaled' (lr_node "
                             "reads the scaled slow features) or 'unscaled' (the normalised ones) — the rule lives in cuicuilco "
                             "@9bfd242, which is not available here, and the two give different features" % method)
        return N.iGSFANode(np.asarray(_get(obj, "x_mean"), dtype=np.float64).reshape(-1), exp, sfa, magn, lr, pca, k,
                           reconstruct_with_sfa=rec and lr is not None, lr_input=igsfa_lr_input or "scaled",
                           scaling=scaling, scaling_matrix=matrix)
    if name in ("Switchboard", "PInvSwitchboard", "Rectangular2dSwitchboard", "Rectangular2dSwitchboardException",
                "DoubleRect2dSwitchboard", "DoubleRhomb2dSwitchboard"):
        i, o = _dims(obj)
        cls = N.PInvSwitchboard if name == "PInvSwitchboard" else N.Switchboard
        sb = cls(i, np.asarray(_get(obj, "connections"), dtype=np.int64).reshape(-1))
        if sb.output_dim != o:
            raise ValueError("Switchboard: %d connections but output_dim %d" % (sb.output_dim, o))
        return sb
    if name == "CloneLayer":
        nodes = _get(obj, "nodes")
        return N.CloneLayer(convert_node(_get(obj, "node", default=nodes[0]), **kw), len(nodes))
    if name == "Layer":
        return N.Layer([convert_node(n, **kw) for n in _get(obj, "nodes")])
    if name == "FlowNode":
        inner = _get(obj, "_flow", "flow")
        return N.FlowNode(convert_flow_object(inner, **kw))
    if name == "IdentityNode":
        return N.IdentityNode(_dims(obj)[0])
    if name == "HeadNode":
        return N.HeadNode(*_dims(obj))
    if name == "CutoffNode":
        return N.CutoffNode(_dims(obj)[0], float(_get(obj, "lower_bound")), float(_get(obj, "upper_bound")))
    raise TypeError("unsupported node class %s.%s — extend pickle_import.convert_node rather than guessing its execute"
                    % (getattr(type(obj), "__module__", "?"), name))


def convert_flow_object(flow_obj, **kw):
    """A pickled mdp.Flow (attribute ``flow``: list of nodes) or a plain list -> list of description nodes."""
    seq = flow_obj if isinstance(flow_obj, (list, tuple)) else _get(flow_obj, "flow")
    out = [convert_node(n, **kw) for n in seq]
    for a, b in zip(out[:-1], out[1:]):
        if a.output_dim != b.input_dim:
            raise ValueError("converted flow: %r -> %r dimension mismatch" % (a, b))
    return out


def load_flow_pickle(path, igsfa_lr_input=None, ignore_attrs=(), pair_prodsadj_reading=None, igsfa_matrix_orientation=None):
    try:
        return convert_flow_object(load_stub_pickle(path), igsfa_lr_input=igsfa_lr_input, ignore_attrs=ignore_attrs,
                                   pair_prodsadj_reading=pair_prodsadj_reading, igsfa_matrix_orientation=igsfa_matrix_orientation)
    except ValueError as e:      # name the file: an integrator converting a directory of SavedNetworks needs to know which one wants which flag
        raise ValueError("%s: %s" % (path, e)) from e


def pickle_to_blob(path, igsfa_lr_input=None, ignore_attrs=(), pair_prodsadj_reading=None, igsfa_matrix_orientation=None):
    return flow_to_blob(load_flow_pickle(path, igsfa_lr_input=igsfa_lr_input, ignore_attrs=ignore_attrs,
                                         pair_prodsadj_reading=pair_prodsadj_reading, igsfa_matrix_orientation=igsfa_matrix_orientation))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="SavedNetworks/<flow>.pckl -> neutral flow blob, without mdp / cuicuilco")
    ap.add_argument("pickle")
    ap.add_argument("out")
    ap.add_argument("--igsfa-lr-input", choices=N.iGSFANode.LR_INPUTS, default=None,
                    help="which slow features the iGSFA linear reconstruction reads (no default: see nodes.iGSFANode)")
    ap.add_argument("--pair-prodsadj-reading", choices=N.PAIR_READINGS, default=None,
                    help="meaning of pair_prodsadj{k}_ex: x_i*x_{i+k} only, or offsets 0..k-1 (no default: see nodes.pair_prodsadj_ex)")
    ap.add_argument("--igsfa-matrix-orientation", choices=MATRIX_ORIENTATIONS, default=None,
                    help="QR_decomposition scaling of iGSFA nodes: s = n @ R.T or s = n @ R (no default: R is square)")
    a = ap.parse_args()
    with open(a.out, "wb") as fh:
        fh.write(pickle_to_blob(a.pickle, igsfa_lr_input=a.igsfa_lr_input, pair_prodsadj_reading=a.pair_prodsadj_reading,
                                igsfa_matrix_orientation=a.igsfa_matrix_orientation))
