"""Flow description objects for the HiGSFA inference path.

These classes mirror, by name and by attribute, the MDP / cuicuilco node objects that
the reference un-pickles from ``SavedNetworks/*.pckl`` and drives through
``networks[k].execute(subimages_arr, benchmark=benchmark)``
(reference: FaceDetectUpdated.py:699, face_analysis.py:1064,1257; class inventory of the
pickles: FaceDetectUpdated.py:57-68; ``Layer.nodes`` / ``IEVMLRecNode.sfa_node``:
face_analysis.py:460-467).

They are *descriptions only*: they carry dimensions and parameters, never arithmetic.
The arithmetic of the product lives in the HIP library behind ``include/higsfa.h``; a
CPU restatement used only as a test oracle lives under ``oracle/``.

Attribute names follow MDP (``avg``, ``v``, ``sf``, ``_bias``, ``connections``, ``nodes``)
and cuicuilco (``funcs``, ``x_mean``, ``sfa_node``, ``pca_node``, ``lr_node``,
``magn_n_sfa_x``, ``num_sfa_features_preserved``) so that a converter from the original
pickles is a one-to-one attribute copy.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "Node", "IdentityNode", "HeadNode", "CutoffNode",
    "PCANode", "WhiteningNode", "SFANode", "GSFANode", "LinearRegressionNode",
    "ExpFunc", "GeneralExpansionNode",
    "identity", "unsigned_08expo", "signed_08expo", "QT", "PAIR_READINGS", "unsigned_expo", "signed_expo", "sel_exp", "pair_prodsadj_ex",
    "iGSFANode", "IEVMLRecNode",
    "Switchboard", "PInvSwitchboard", "Rectangular2dSwitchboard",
    "Layer", "CloneLayer", "FlowNode",
]


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


class Node(object):
    """Base of every flow element: maps (N, input_dim) -> (N, output_dim)."""

    input_dim = None
    output_dim = None

    def is_trainable(self):
        return False

    def is_training(self):
        return False

    def __repr__(self):
        return "%s(input_dim=%s, output_dim=%s)" % (
            type(self).__name__, self.input_dim, self.output_dim)


class IdentityNode(Node):
    """mdp.nodes.IdentityNode: y = x."""

    def __init__(self, input_dim):
        self.input_dim = int(input_dim)
        self.output_dim = int(input_dim)


class HeadNode(Node):
    """cuicuilco.more_nodes.HeadNode: y = x[:, :output_dim]."""

    def __init__(self, input_dim, output_dim):
        self.input_dim = int(input_dim)
        self.output_dim = int(output_dim)
        if not 0 < self.output_dim <= self.input_dim:
            raise ValueError("HeadNode: need 0 < output_dim <= input_dim")


class CutoffNode(Node):
    """mdp.nodes.CutoffNode: y = clip(x, lower_bound, upper_bound)."""

    def __init__(self, input_dim, lower_bound, upper_bound):
        self.input_dim = int(input_dim)
        self.output_dim = int(input_dim)
        self.lower_bound = float(lower_bound)
        self.upper_bound = float(upper_bound)


class PCANode(Node):
    """mdp.nodes.PCANode after training: ``y = (x - avg) @ v``.

    avg : (1, input_dim)   v : (input_dim, output_dim)
    """

    def __init__(self, avg, v):
        v = _f64(v)
        self.v = v
        self.input_dim, self.output_dim = int(v.shape[0]), int(v.shape[1])
        self.avg = _f64(avg, (1, self.input_dim))


class WhiteningNode(PCANode):
    """mdp.nodes.WhiteningNode: PCANode whose columns of ``v`` are scaled by 1/sqrt(eigenvalue)."""


class SFANode(Node):
    """mdp.nodes.SFANode after training: ``y = x @ sf - _bias`` with ``_bias = avg @ sf``
    (SURVEY.md §8a row a7).  Columns are ordered slowest first; callers consume the first k
    (FaceDetectUpdated.py:719).

    avg : (1, input_dim)   sf : (input_dim, output_dim)   _bias : (1, output_dim)
    """

    def __init__(self, avg, sf, bias=None):
        sf = _f64(sf)
        self.sf = sf
        self.input_dim, self.output_dim = int(sf.shape[0]), int(sf.shape[1])
        self.avg = _f64(avg, (1, self.input_dim))
        if bias is None:
            bias = self.avg @ self.sf
        self._bias = _f64(bias, (1, self.output_dim))


class GSFANode(SFANode):
    """cuicuilco.gsfa_node.GSFANode: same execute as SFANode (graph only changes training)."""


class LinearRegressionNode(Node):
    """mdp.nodes.LinearRegressionNode (with_bias=True): ``y = [1, x] @ beta``.

    beta : (input_dim + 1, output_dim); row 0 is the intercept.
    """

    def __init__(self, beta):
        beta = _f64(beta)
        self.beta = beta
        self.input_dim, self.output_dim = int(beta.shape[0]) - 1, int(beta.shape[1])


class ExpFunc(object):
    """One entry of ``GeneralExpansionNode.funcs`` (cuicuilco.nonlinear_expansion vocabulary,
    SURVEY.md §8a row a6).  ``sel`` > 0 restricts the function to the first ``sel`` input
    columns (cuicuilco's ``sel_exp(n, func)``).

    kind        output for an input block x of d columns
    ----------  ------------------------------------------------------------
    identity    x                                            (d columns)
    abs_pow     abs(x) ** expo                               (d)
    signed_pow  sign(x) * abs(x) ** expo                     (d)
    quadratic   x_i * x_j for i <= j, i-major                (d (d + 1) / 2)
    pair_adj    x_i * x_{i+k} for i in range(d - k)          (d - k)
    pair_band   hstack over off = 0 .. k-1 of x_i * x_{i+off} (squares first)   (sum of d - off)

    ``pair_adj`` / ``pair_band`` are the two readings of cuicuilco's ``pair_prodsadj{k}_ex`` family that public
    descriptions allow (is ``k`` ONE offset, or the number of offsets 0 .. k-1 of a reflexive band?); neither can be
    checked in this repository, so the name alone never selects one — see ``pair_prodsadj_ex``.
    """

    KINDS = ("identity", "abs_pow", "signed_pow", "quadratic", "pair_adj", "pair_band")

    def __init__(self, kind, expo=1.0, k=0, sel=0, name=None):
        if kind not in self.KINDS:
            raise ValueError("unknown expansion kind %r" % (kind,))
        self.kind = kind
        self.expo = float(expo)
        self.k = int(k)
        self.sel = int(sel)
        self.__name__ = name or kind

    def used_dim(self, input_dim):
        return min(self.sel, input_dim) if self.sel > 0 else input_dim

    def out_dim(self, input_dim):
        d = self.used_dim(input_dim)
        if self.kind in ("identity", "abs_pow", "signed_pow"):
            return d
        if self.kind == "quadratic":
            return d * (d + 1) // 2
        if self.kind == "pair_band":
            return sum(max(d - off, 0) for off in range(self.k))
        return max(d - self.k, 0)

    def __repr__(self):
        return "ExpFunc(%s)" % self.__name__


identity = ExpFunc("identity", name="identity")
unsigned_08expo = ExpFunc("abs_pow", expo=0.8, name="unsigned_08expo")
signed_08expo = ExpFunc("signed_pow", expo=0.8, name="signed_08expo")
QT = ExpFunc("quadratic", name="QT")
PAIR_READINGS = ("offset", "band")


def unsigned_expo(expo):
    return ExpFunc("abs_pow", expo=expo, name="unsigned_expo(%g)" % expo)


def signed_expo(expo):
    return ExpFunc("signed_pow", expo=expo, name="signed_expo(%g)" % expo)


def pair_prodsadj_ex(k, reading):
    """cuicuilco.nonlinear_expansion ``pair_prodsadj{k}_ex`` under an EXPLICIT reading (no default, like
    ``iGSFANode.lr_input``):

    "offset"  products of columns exactly k apart:   x_i * x_{i+k}                      (d - k columns)
    "band"    reflexive band of k offsets 0 .. k-1:  [x_i * x_i, x_i * x_{i+1}, ...]    (k d - k (k - 1) / 2 columns)
    """
    if reading not in PAIR_READINGS:
        raise ValueError("pair_prodsadj%d_ex: state reading='offset' (x_i * x_{i+%d} only) or reading='band' (offsets 0..%d, "
                         "squares included); the name does not decide it" % (k, k, k - 1))
    if reading == "offset":
        return ExpFunc("pair_adj", k=k, name="pair_prodsadj%d_ex[offset]" % k)
    return ExpFunc("pair_band", k=k, name="pair_prodsadj%d_ex[band]" % k)


def sel_exp(n, func):
    return ExpFunc(func.kind, expo=func.expo, k=func.k, sel=n,
                   name="sel_exp(%d,%s)" % (n, func.__name__))


class GeneralExpansionNode(Node):
    """cuicuilco.more_nodes.GeneralExpansionNode: ``y = hstack([f(x) for f in funcs])``."""

    def __init__(self, funcs, input_dim):
        self.funcs = list(funcs)
        self.input_dim = int(input_dim)
        self.output_dim = int(sum(f.out_dim(self.input_dim) for f in self.funcs))


class iGSFANode(Node):
    """cuicuilco.igsfa_node.iGSFANode (older pickles: IEVMLRecNode, face_analysis.py:463).

    Execute semantics restated from the published HiGSFA description (arXiv:1601.03945) and
    SURVEY.md §8a row a8 — source not available in this container, so this spec is owned by
    the build and the two points where public descriptions of cuicuilco differ are EXPLICIT
    fields of the node (and of its blob record), never an implicit choice:

        x0  = x - x_mean
        e   = exp_node(x0)                      (or x0 when exp_node is None)
        n   = sfa_node(e)                       (normalised slow features)
        s   = n * magn_n_sfa_x                  scaling == "per_column"  (sensitivity_based / data_dependent)
            = n @ scaling_matrix                scaling == "matrix"      (QR_decomposition: scaling_matrix = R.T)
        r   = x0 - lr_node(s)                   lr_input == "scaled"     (round-1 reading)
            = x0 - lr_node(n)                   lr_input == "unscaled"   (reconstruction trained on the normalised features)
            = x0                                when not reconstruct_with_sfa
        q   = pca_node(r)
        y   = hstack([s[:, :num_sfa_features_preserved], q])
    """

    LR_INPUTS = ("scaled", "unscaled")
    SCALINGS = ("per_column", "matrix")

    def __init__(self, x_mean, exp_node, sfa_node, magn_n_sfa_x, lr_node, pca_node,
                 num_sfa_features_preserved, reconstruct_with_sfa=True, lr_input="scaled",
                 scaling="per_column", scaling_matrix=None):
        if lr_input not in self.LR_INPUTS:
            raise ValueError("iGSFANode: lr_input must be one of %r" % (self.LR_INPUTS,))
        if scaling not in self.SCALINGS:
            raise ValueError("iGSFANode: scaling must be one of %r" % (self.SCALINGS,))
        self.sfa_node = sfa_node
        self.pca_node = pca_node
        self.exp_node = exp_node
        self.lr_node = lr_node if reconstruct_with_sfa else None
        self.reconstruct_with_sfa = bool(reconstruct_with_sfa)
        self.lr_input = lr_input
        self.scaling = scaling
        self.input_dim = int(pca_node.input_dim)
        self.x_mean = _f64(x_mean, (1, self.input_dim))
        k = sfa_node.output_dim
        if scaling == "matrix":
            if scaling_matrix is None:
                raise ValueError("iGSFANode: scaling='matrix' needs scaling_matrix")
            self.scaling_matrix = _f64(scaling_matrix, (k, k))
            self.magn_n_sfa_x = np.ones((1, k))
        else:
            if scaling_matrix is not None:
                raise ValueError("iGSFANode: scaling_matrix given with scaling='per_column'")
            self.scaling_matrix = None
            self.magn_n_sfa_x = _f64(magn_n_sfa_x, (1, k))
        self.num_sfa_features_preserved = int(num_sfa_features_preserved)
        if self.num_sfa_features_preserved > sfa_node.output_dim:
            raise ValueError("iGSFANode: num_sfa_features_preserved > sfa_node.output_dim")
        e_dim = exp_node.output_dim if exp_node is not None else self.input_dim
        if exp_node is not None and exp_node.input_dim != self.input_dim:
            raise ValueError("iGSFANode: exp_node.input_dim mismatch")
        if sfa_node.input_dim != e_dim:
            raise ValueError("iGSFANode: sfa_node.input_dim != expanded dim")
        if self.lr_node is not None and (self.lr_node.input_dim != sfa_node.output_dim
                                         or self.lr_node.output_dim != self.input_dim):
            raise ValueError("iGSFANode: lr_node dims mismatch")
        self.output_dim = self.num_sfa_features_preserved + int(pca_node.output_dim)


IEVMLRecNode = iGSFANode


class Switchboard(Node):
    """mdp.hinet.Switchboard: ``y = x[:, connections]``."""

    def __init__(self, input_dim, connections):
        self.input_dim = int(input_dim)
        self.connections = np.ascontiguousarray(np.asarray(connections, dtype=np.int64))
        if self.connections.ndim != 1:
            raise ValueError("Switchboard: connections must be 1-d")
        if self.connections.size and (self.connections.min() < 0
                                      or self.connections.max() >= self.input_dim):
            raise ValueError("Switchboard: connection index out of range")
        self.output_dim = int(self.connections.size)


class PInvSwitchboard(Switchboard):
    """cuicuilco.more_nodes.PInvSwitchboard: Switchboard with a pseudo-inverse (unused at execute)."""


class Rectangular2dSwitchboard(Switchboard):
    """mdp.hinet.Rectangular2dSwitchboard: rectangular receptive fields over a 2-d grid of
    channels.  Connection order: field row (y) major, then field
    column, then row inside the field, then column inside the field, then channel:

        conn = ((fy*sy + py) * W_in + (fx*sx + px)) * C + c
    """

    def __init__(self, in_channels_xy, field_channels_xy, field_spacing_xy=None,
                 in_channel_dim=1):
        wx, wy = int(in_channels_xy[0]), int(in_channels_xy[1])
        fx_, fy_ = int(field_channels_xy[0]), int(field_channels_xy[1])
        if field_spacing_xy is None:
            field_spacing_xy = (fx_, fy_)
        sx, sy = int(field_spacing_xy[0]), int(field_spacing_xy[1])
        c = int(in_channel_dim)
        if (wx - fx_) % sx or (wy - fy_) % sy:
            raise ValueError("Rectangular2dSwitchboard: fields do not tile the input")
        nx, ny = (wx - fx_) // sx + 1, (wy - fy_) // sy + 1
        fy, fx, py, px, cc = np.meshgrid(np.arange(ny), np.arange(nx), np.arange(fy_),
                                         np.arange(fx_), np.arange(c), indexing="ij")
        conn = ((fy * sy + py) * wx + (fx * sx + px)) * c + cc
        super(Rectangular2dSwitchboard, self).__init__(wx * wy * c, conn.reshape(-1))
        self.in_channels_xy = (wx, wy)
        self.field_channels_xy = (fx_, fy_)
        self.field_spacing_xy = (sx, sy)
        self.in_channel_dim = c
        self.out_channels_xy = (nx, ny)
        self.out_channel_dim = fx_ * fy_ * c
        self.output_channels = nx * ny


class Layer(Node):
    """mdp.hinet.Layer: node k reads the next ``nodes[k].input_dim`` input columns and writes
    the next ``nodes[k].output_dim`` output columns."""

    def __init__(self, nodes):
        self.nodes = list(nodes)
        if not self.nodes:
            raise ValueError("Layer: needs at least one node")
        self.input_dim = int(sum(n.input_dim for n in self.nodes))
        self.output_dim = int(sum(n.output_dim for n in self.nodes))

    def __len__(self):
        return len(self.nodes)

    def __getitem__(self, i):
        return self.nodes[i]


class CloneLayer(Layer):
    """mdp.hinet.CloneLayer: the same node object applied to ``n_nodes`` equal slices."""

    def __init__(self, node, n_nodes=1):
        self.node = node
        super(CloneLayer, self).__init__([node] * int(n_nodes))


class FlowNode(Node):
    """mdp.hinet.FlowNode: a sequence of nodes used as one node (``.flow`` is the list)."""

    def __init__(self, flow):
        self.flow = list(flow.flow) if hasattr(flow, "flow") else list(flow)
        if not self.flow:
            raise ValueError("FlowNode: empty flow")
        for a, b in zip(self.flow[:-1], self.flow[1:]):
            if a.output_dim != b.input_dim:
                raise ValueError("FlowNode: dimension mismatch %s -> %s" % (a, b))
        self.input_dim = int(self.flow[0].input_dim)
        self.output_dim = int(self.flow[-1].output_dim)

    def __len__(self):
        return len(self.flow)

    def __getitem__(self, i):
        return self.flow[i]
