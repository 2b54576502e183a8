"""Offline converter: pickled MDP / cuicuilco flow  ->  description objects  ->  neutral blob,
WITHOUT importing mdp or cuicuilco.

The reference loads its networks with ``cache_obj.load_obj_from_cache`` = ``pickle.load`` of an
``mdp.Flow`` object graph (face_analysis.py:457, 473-478), which needs mdp-toolkit, cuicuilco and
the module aliases of FaceDetectUpdated.py:57-68 on the path.  Here the pickle is read with
``classifier.StubUnpickler``: every non-numpy global becomes an attribute bag that remembers its
module and class name, and this module maps those bags, by class NAME and MDP attribute names,
onto ``pyfaceanalysis_amd.nodes``.  Unknown node classes or expansion functions raise — nothing
is guessed.  The trained flows themselves are not shipped with the reference
(.MISSING_LARGE_BLOBS), so this is exercised on synthetic pickles built under the same module /
class names (tests/test_pickle_import.py).
"""
from __future__ import annotations

import numpy as np

from . import nodes as N
from .blob import flow_to_blob
from .classifier import load_stub_pickle

# cuicuilco.nonlinear_expansion function names -> ExpFunc  
_FUNCS = {
    "identity": N.identity, "I": N.identity,
    "unsigned_08expo": N.unsigned_08expo, "signed_08expo": N.signed_08expo,
    "QT": N.QT, "pair_prodsadj1_ex": N.pair_prodsadj1_ex, "pair_prodsadj2_ex": N.pair_prodsadj2_ex,
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


def convert_func(f):
    name = getattr(f, "__name__", None) or _cls(f)
    if name not in _FUNCS:
        raise TypeError("unsupported expansion function %r (module %r)" % (name, getattr(f, "__module__", "?")))
    return _FUNCS[name]


def convert_node(obj):
    """One pickled node (stub) -> a pyfaceanalysis_amd.nodes object."""
    if obj is None:
        return None
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
        return N.LinearRegressionNode(np.asarray(_get(obj, "beta"), dtype=np.float64))
    if name == "GeneralExpansionNode":
        i, o = _dims(obj)
        node = N.GeneralExpansionNode([convert_func(f) for f in _get(obj, "funcs")], i)
        if node.output_dim != o:
            raise ValueError("GeneralExpansionNode: converted width %d != pickled output_dim %d" % (node.output_dim, o))
        return node
    if name in ("iGSFANode", "IEVMLRecNode"):
        exp = convert_node(_get(obj, "exp_node", default=None))
        sfa = convert_node(_get(obj, "sfa_node"))
        pca = convert_node(_get(obj, "pca_node"))
        rec = bool(_get(obj, "reconstruct_with_sfa", default=True))
        lr = convert_node(_get(obj, "lr_node", default=None)) if rec else None
        magn = np.asarray(_get(obj, "magn_n_sfa_x", default=np.ones(sfa.output_dim)), dtype=np.float64).reshape(-1)
        k = int(_get(obj, "num_sfa_features_preserved", default=sfa.output_dim))
        return N.iGSFANode(np.asarray(_get(obj, "x_mean"), dtype=np.float64).reshape(-1), exp, sfa, magn, lr, pca, k,
                           reconstruct_with_sfa=rec and lr is not None)
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
        return N.CloneLayer(convert_node(_get(obj, "node", default=nodes[0])), len(nodes))
    if name == "Layer":
        return N.Layer([convert_node(n) for n in _get(obj, "nodes")])
    if name == "FlowNode":
        inner = _get(obj, "_flow", "flow")
        return N.FlowNode(convert_flow_object(inner))
    if name == "IdentityNode":
        return N.IdentityNode(_dims(obj)[0])
    if name == "HeadNode":
        return N.HeadNode(*_dims(obj))
    if name == "CutoffNode":
        return N.CutoffNode(_dims(obj)[0], float(_get(obj, "lower_bound")), float(_get(obj, "upper_bound")))
    raise TypeError("unsupported node class %s.%s — extend pickle_import.convert_node rather than guessing its execute"
                    % (getattr(type(obj), "__module__", "?"), name))


def convert_flow_object(flow_obj):
    """A pickled mdp.Flow (attribute ``flow``: list of nodes) or a plain list -> list of description nodes."""
    seq = flow_obj if isinstance(flow_obj, (list, tuple)) else _get(flow_obj, "flow")
    out = [convert_node(n) for n in seq]
    for a, b in zip(out[:-1], out[1:]):
        if a.output_dim != b.input_dim:
            raise ValueError("converted flow: %r -> %r dimension mismatch" % (a, b))
    return out


def load_flow_pickle(path):
    return convert_flow_object(load_stub_pickle(path))


def pickle_to_blob(path):
    return flow_to_blob(load_flow_pickle(path))


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 3:
        sys.exit("usage: python -m pyfaceanalysis_amd.pickle_import SavedNetworks/<flow>.pckl out.hgflow")
    with open(sys.argv[2], "wb") as fh:
        fh.write(pickle_to_blob(sys.argv[1]))
