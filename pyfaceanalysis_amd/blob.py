"""Neutral flow blob ("HGSFAFL1"): the model-input side of the drop-in boundary.

The reference hands its flows around as Python pickles of MDP object graphs
(face_analysis.py:457, ``cache_obj.load_obj_from_cache``); a C ABI cannot consume those.
This module serialises a tree of :mod:`pyfaceanalysis_amd.nodes` objects into one
little-endian byte string that ``hg_flow_load`` (include/higsfa.h) parses, and reads it
back (round-trip tests, tools).  Every record and every array starts 8-byte aligned.

    header : char[8] "HGSFAFL1" | u32 version=1 | u32 flags=0 | u64 total_bytes
    node   : u32 kind | u32 in_dim | u32 out_dim | u32 aux | payload
    kinds  : 1 FLOW(aux=n; n nodes)            2 SWITCHBOARD(aux=n; i32[n] connections)
             3 LAYER(aux=n; n nodes)           4 CLONELAYER(aux=copies; 1 node)
             5 AFFINE(aux=subtype; f64 a[in], f64 W[in*out] row-major, f64 b[out])
                                               y = (x - a) @ W + b
             6 EXPANSION(aux=n; n x {u32 kind,u32 sel,u32 k,u32 0,f64 expo})
             7 IGSFA(aux=num_sfa_features_preserved; u32 has_exp,u32 flags; f64 x_mean[in];
                     [EXPANSION]; AFFINE sfa; f64 magn[sfa.out] or f64 M[sfa.out^2]; [AFFINE lr]; AFFINE pca)
                     flags: bit0 lr_node present; bit1 lr_node reads the UNSCALED slow features
                     (lr_input="unscaled"); bit2 scaling is a matrix M (s = n @ M, row-major) instead
                     of a per-column scale
             8 IDENTITY   9 HEAD   10 CUTOFF(f64 lo, f64 hi)   11 FLOWNODE(aux=n; n nodes)
"""
from __future__ import annotations

import struct

import numpy as np

from . import nodes as N

MAGIC = b"HGSFAFL1"
VERSION = 1

K_FLOW, K_SWITCHBOARD, K_LAYER, K_CLONELAYER, K_AFFINE, K_EXPANSION = 1, 2, 3, 4, 5, 6
K_IGSFA, K_IDENTITY, K_HEAD, K_CUTOFF, K_FLOWNODE = 7, 8, 9, 10, 11

AFF_GENERIC, AFF_PCA, AFF_WHITENING, AFF_SFA, AFF_GSFA, AFF_LINREG = 0, 1, 2, 3, 4, 5
IG_HAS_LR, IG_LR_UNSCALED, IG_SCALE_MATRIX = 1, 2, 4
_EXP_KIND = {"identity": 0, "abs_pow": 1, "signed_pow": 2, "quadratic": 3, "pair_adj": 4, "pair_band": 5}
_EXP_NAME = {v: k for k, v in _EXP_KIND.items()}


class _Writer(object):
    def __init__(self):
        self.parts = []
        self.n = 0

    def raw(self, b):
        self.parts.append(b)
        self.n += len(b)
        pad = (-self.n) % 8
        if pad:
            self.parts.append(b"\0" * pad)
            self.n += pad

    def head(self, kind, in_dim, out_dim, aux):
        self.raw(struct.pack("<IIII", kind, in_dim, out_dim, aux))

    def f64(self, a, count):
        a = np.ascontiguousarray(np.asarray(a, dtype="<f8")).reshape(-1)
        if a.size != count:
            raise ValueError("blob: array has %d elements, expected %d" % (a.size, count))
        self.raw(a.tobytes())

    def i32(self, a):
        self.raw(np.ascontiguousarray(np.asarray(a, dtype="<i4")).tobytes())


def _affine_params(node):
    """(subtype, a, W, b) with y = (x - a) @ W + b for every affine MDP node."""
    i, o = node.input_dim, node.output_dim
    if isinstance(node, N.PCANode):
        sub = AFF_WHITENING if isinstance(node, N.WhiteningNode) else AFF_PCA
        return sub, node.avg.reshape(-1), node.v, np.zeros(o)
    if isinstance(node, N.SFANode):
        sub = AFF_GSFA if isinstance(node, N.GSFANode) else AFF_SFA
        # MDP: mult(x, sf) - _bias  (avg is folded into _bias by MDP itself)
        return sub, np.zeros(i), node.sf, -node._bias.reshape(-1)
    if isinstance(node, N.LinearRegressionNode):
        return AFF_LINREG, np.zeros(i), node.beta[1:], node.beta[0]
    raise TypeError("not an affine node: %r" % (node,))


def _write_node(w, node):
    i, o = node.input_dim, node.output_dim
    if isinstance(node, (N.PCANode, N.SFANode, N.LinearRegressionNode)):
        sub, a, W, b = _affine_params(node)
        w.head(K_AFFINE, i, o, sub)
        w.f64(a, i)
        w.f64(W, i * o)
        w.f64(b, o)
    elif isinstance(node, N.Switchboard):
        w.head(K_SWITCHBOARD, i, o, o)
        w.i32(node.connections)
    elif isinstance(node, N.CloneLayer):
        w.head(K_CLONELAYER, i, o, len(node.nodes))
        _write_node(w, node.node)
    elif isinstance(node, N.Layer):
        w.head(K_LAYER, i, o, len(node.nodes))
        for c in node.nodes:
            _write_node(w, c)
    elif isinstance(node, N.FlowNode):
        w.head(K_FLOWNODE, i, o, len(node.flow))
        for c in node.flow:
            _write_node(w, c)
    elif isinstance(node, N.GeneralExpansionNode):
        w.head(K_EXPANSION, i, o, len(node.funcs))
        for f in node.funcs:
            w.raw(struct.pack("<IIIId", _EXP_KIND[f.kind], f.sel, f.k, 0, f.expo))
    elif isinstance(node, N.iGSFANode):
        w.head(K_IGSFA, i, o, node.num_sfa_features_preserved)
        flags = (IG_HAS_LR if node.lr_node is not None else 0) | (IG_LR_UNSCALED if node.lr_input == "unscaled" else 0) \
            | (IG_SCALE_MATRIX if node.scaling == "matrix" else 0)
        w.raw(struct.pack("<II", 1 if node.exp_node is not None else 0, flags))
        w.f64(node.x_mean, i)
        if node.exp_node is not None:
            _write_node(w, node.exp_node)
        _write_node(w, node.sfa_node)
        if node.scaling == "matrix":
            w.f64(node.scaling_matrix, node.sfa_node.output_dim ** 2)
        else:
            w.f64(node.magn_n_sfa_x, node.sfa_node.output_dim)
        if node.lr_node is not None:
            _write_node(w, node.lr_node)
        _write_node(w, node.pca_node)
    elif isinstance(node, N.IdentityNode):
        w.head(K_IDENTITY, i, o, 0)
    elif isinstance(node, N.HeadNode):
        w.head(K_HEAD, i, o, 0)
    elif isinstance(node, N.CutoffNode):
        w.head(K_CUTOFF, i, o, 0)
        w.raw(struct.pack("<dd", node.lower_bound, node.upper_bound))
    else:
        raise TypeError("blob: unsupported node class %s (fail loudly rather than guess)"
                        % type(node).__name__)


def flow_to_blob(flow_nodes):
    """Serialise a list of top-level nodes (``Flow.flow``) into one ``bytes`` object."""
    flow_nodes = list(flow_nodes)
    if not flow_nodes:
        raise ValueError("blob: empty flow")
    for a, b in zip(flow_nodes[:-1], flow_nodes[1:]):
        if a.output_dim != b.input_dim:
            raise ValueError("blob: dimension mismatch %r -> %r" % (a, b))
    w = _Writer()
    w.raw(MAGIC + struct.pack("<IIQ", VERSION, 0, 0))
    w.head(K_FLOW, flow_nodes[0].input_dim, flow_nodes[-1].output_dim, len(flow_nodes))
    for n in flow_nodes:
        _write_node(w, n)
    out = bytearray(b"".join(w.parts))
    struct.pack_into("<Q", out, 16, len(out))
    return bytes(out)


class _Reader(object):
    def __init__(self, buf):
        self.buf = memoryview(buf)
        self.p = 0

    def take(self, n):
        if self.p + n > len(self.buf):
            raise ValueError("blob: truncated")
        b = self.buf[self.p:self.p + n]
        self.p += n + ((-n) % 8)
        return b

    def head(self):
        return struct.unpack("<IIII", self.take(16))

    def f64(self, count):
        return np.frombuffer(self.take(8 * count), dtype="<f8").astype(np.float64)


def _read_node(r):
    kind, i, o, aux = r.head()
    if kind == K_AFFINE:
        a, W, b = r.f64(i), r.f64(i * o).reshape(i, o), r.f64(o)
        if aux in (AFF_PCA, AFF_WHITENING):
            cls = N.WhiteningNode if aux == AFF_WHITENING else N.PCANode
            if np.any(b != 0):
                raise ValueError("blob: PCA record with non-zero post-bias")
            return cls(a, W)
        if aux in (AFF_SFA, AFF_GSFA):
            cls = N.GSFANode if aux == AFF_GSFA else N.SFANode
            return cls(np.zeros(i), W, -b)
        if aux == AFF_LINREG:
            return N.LinearRegressionNode(np.vstack([b.reshape(1, o), W]))
        raise ValueError("blob: unknown affine subtype %d" % aux)
    if kind == K_SWITCHBOARD:
        conn = np.frombuffer(r.take(4 * aux), dtype="<i4").astype(np.int64)
        return N.Switchboard(i, conn)
    if kind == K_CLONELAYER:
        return N.CloneLayer(_read_node(r), aux)
    if kind == K_LAYER:
        return N.Layer([_read_node(r) for _ in range(aux)])
    if kind == K_FLOWNODE:
        return N.FlowNode([_read_node(r) for _ in range(aux)])
    if kind == K_EXPANSION:
        funcs = []
        for _ in range(aux):
            k, sel, kk, _z, expo = struct.unpack("<IIIId", r.take(24))
            funcs.append(N.ExpFunc(_EXP_NAME[k], expo=expo, k=kk, sel=sel))
        return N.GeneralExpansionNode(funcs, i)
    if kind == K_IGSFA:
        has_exp, flags = struct.unpack("<II", r.take(8))
        if flags & ~(IG_HAS_LR | IG_LR_UNSCALED | IG_SCALE_MATRIX):
            raise ValueError("blob: unknown iGSFA flags 0x%x" % flags)
        x_mean = r.f64(i)
        exp_node = _read_node(r) if has_exp else None
        sfa = _read_node(r)
        k = sfa.output_dim
        matrix = r.f64(k * k).reshape(k, k) if flags & IG_SCALE_MATRIX else None
        magn = None if flags & IG_SCALE_MATRIX else r.f64(k)
        lr = _read_node(r) if flags & IG_HAS_LR else None
        pca = _read_node(r)
        return N.iGSFANode(x_mean, exp_node, sfa, magn, lr, pca, aux, reconstruct_with_sfa=bool(flags & IG_HAS_LR),
                           lr_input="unscaled" if flags & IG_LR_UNSCALED else "scaled",
                           scaling="matrix" if flags & IG_SCALE_MATRIX else "per_column", scaling_matrix=matrix)
    if kind == K_IDENTITY:
        return N.IdentityNode(i)
    if kind == K_HEAD:
        return N.HeadNode(i, o)
    if kind == K_CUTOFF:
        lo, hi = struct.unpack("<dd", r.take(16))
        return N.CutoffNode(i, lo, hi)
    raise ValueError("blob: unknown node kind %d" % kind)


def blob_to_flow(blob):
    """Parse a blob back into a list of top-level node objects."""
    r = _Reader(blob)
    hdr = r.take(24)
    if bytes(hdr[:8]) != MAGIC:
        raise ValueError("blob: bad magic")
    version, _flags, total = struct.unpack("<IIQ", hdr[8:24])
    if version != VERSION:
        raise ValueError("blob: unsupported version %d" % version)
    if total != len(blob):
        raise ValueError("blob: size field %d != buffer size %d" % (total, len(blob)))
    kind, _i, _o, n = r.head()
    if kind != K_FLOW:
        raise ValueError("blob: root record is not a FLOW")
    return [_read_node(r) for _ in range(n)]
