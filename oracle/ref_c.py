"""TEST ORACLE — NOT PRODUCT CODE.  *** parity unpinned *** (see mdp_restate.py header).

Driver of the C restatement (oracle/ref_c.c): walks a flow exactly like mdp.Flow / mdp.hinet.Layer
do and calls the C leaf functions through ctypes.  Exists so that the
numpy restatement is cross-checked by an implementation that shares no arithmetic code with it
(different language, no BLAS, different summation order).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libref_c.so")
        src = os.path.join(_HERE, "ref_c.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(so)
        dp, lp, l, d = C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_long, C.c_double
        L.ref_gather_f64.argtypes = [dp, l, l, lp, l, dp, l]
        L.ref_affine_f64.argtypes = [dp, l, l, l, dp, dp, dp, l, dp, l]
        L.ref_expfunc_f64.argtypes = [dp, l, l, l, C.c_int, d, l, dp, l]
        L.ref_expfunc_f64.restype = l
        L.ref_gauss_regression_f64.argtypes = [dp, l, l, l, l, dp, dp, dp, dp, dp, dp, dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _names(obj):
    return [c.__name__ for c in type(obj).__mro__]


_KIND = {"identity": 0, "abs_pow": 1, "signed_pow": 2, "quadratic": 3, "pair_adj": 4, "pair_band": 5}


def _affine(x, a, W, b):
    x, a, W, b = _c(x), _c(a).reshape(-1), _c(W), _c(b).reshape(-1)
    y = np.empty((x.shape[0], W.shape[1]))
    lib().ref_affine_f64(_p(x), x.shape[0], x.shape[1], W.shape[0], _p(a), _p(W), _p(b), W.shape[1], _p(y), y.shape[1])
    return y


def execute_node(node, x):
    x = _c(x)
    if x.shape[1] != node.input_dim:
        raise ValueError("ref_c: dimension mismatch at %s" % type(node).__name__)
    names = _names(node)
    n = x.shape[0]
    if "iGSFANode" in names:
        x0 = x - node.x_mean
        e = execute_node(node.exp_node, x0) if node.exp_node is not None else x0
        nsf = execute_node(node.sfa_node, e)
        if getattr(node, "scaling", "per_column") == "matrix":       # s = n @ M through the C affine (a = 0, b = 0)
            k = nsf.shape[1]
            s = _affine(nsf, np.zeros(k), node.scaling_matrix, np.zeros(k))
        else:
            s = nsf * node.magn_n_sfa_x
        lr_in = nsf if getattr(node, "lr_input", "scaled") == "unscaled" else s
        r = x0 - execute_node(node.lr_node, lr_in) if node.lr_node is not None else x0
        q = execute_node(node.pca_node, r)
        return np.concatenate([s[:, :node.num_sfa_features_preserved], q], axis=1)
    if "PCANode" in names:
        return _affine(x, node.avg, node.v, np.zeros(node.output_dim))
    if "SFANode" in names:
        return _affine(x, np.zeros(node.input_dim), node.sf, -node._bias)
    if "LinearRegressionNode" in names:
        return _affine(x, np.zeros(node.input_dim), node.beta[1:], node.beta[0])
    if "GeneralExpansionNode" in names:
        y = np.empty((n, node.output_dim))
        o = 0
        for f in node.funcs:
            d = min(f.sel, node.input_dim) if f.sel > 0 else node.input_dim
            m = lib().ref_expfunc_f64(_p(x), n, x.shape[1], d, _KIND[f.kind], f.expo, f.k,
                                      C.cast(C.c_void_p(y.ctypes.data + 8 * o), C.POINTER(C.c_double)), y.shape[1])
            o += m
        assert o == node.output_dim
        return y
    if "Switchboard" in names:
        idx = np.ascontiguousarray(node.connections, dtype=np.int64)
        y = np.empty((n, idx.size))
        lib().ref_gather_f64(_p(x), n, x.shape[1], idx.ctypes.data_as(C.POINTER(C.c_long)), idx.size, _p(y), idx.size)
        return y
    if "Layer" in names:
        y = np.empty((n, node.output_dim))
        i0 = o0 = 0
        for sub in node.nodes:
            y[:, o0:o0 + sub.output_dim] = execute_node(sub, x[:, i0:i0 + sub.input_dim])
            i0 += sub.input_dim
            o0 += sub.output_dim
        return y
    if "FlowNode" in names:
        for sub in node.flow:
            x = execute_node(sub, x)
        return x
    if "IdentityNode" in names:
        return x
    if "HeadNode" in names:
        return x[:, :node.output_dim].copy()
    if "CutoffNode" in names:
        return np.clip(x, node.lower_bound, node.upper_bound)
    raise TypeError("ref_c: no restatement for %s" % type(node).__name__)


def execute_flow(flow_nodes, x):
    x = np.asarray(x, dtype=np.float64)
    for node in flow_nodes:
        x = execute_node(node, x)
    return x


def gauss_regression(x, means, inv_covs, sqrt_det, prior, avg_labels, want_std=True):
    x = _c(x)
    K, d = means.shape
    reg = np.empty(x.shape[0])
    sd = np.empty(x.shape[0]) if want_std else None
    lib().ref_gauss_regression_f64(_p(x), x.shape[0], x.shape[1], K, d, _p(_c(means)), _p(_c(inv_covs)),
                                   _p(_c(sqrt_det)), _p(_c(prior)), _p(_c(avg_labels)), _p(reg),
                                   _p(sd) if want_std else None)
    return (reg, sd) if want_std else reg
