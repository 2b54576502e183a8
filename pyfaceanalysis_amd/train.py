"""SFA training step on the GPU (BASELINE.json configs[4]): per-node covariance
accumulation (HIP, fp64) + batched generalised eigenproblem (rocSOLVER) through ``hg_sfa_train_layer``.
Not on the reference's path (PyFaceAnalysis never trains); provided as the counterpart of
mdp.nodes.SFANode.train / stop_training for one layer of nodes."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def sfa_train_layer(x, conn, device=0, x_dtype=None, n=None, ldx=None):
    """x: (n, ldx) matrix in time order — a numpy array (copied to the GPU) or a raw device pointer (int,
    e.g. ``tensor.data_ptr()``, with x_dtype / n / ldx given);
    conn: (n_nodes, d) int array of input columns per node.  Returns (evals (n_nodes, d),
    evecs (n_nodes, d, d) with evecs[k][:, i] = i-th eigenvector (slowest first, w' B w = 1),
    mean (n_nodes, d), (stats_ms, solve_ms))."""
    on_host = 0
    if isinstance(x, np.ndarray):
        if x.ndim != 2:
            raise ValueError("x must be 2-d")
        if _capi.np_dtype_code(x.dtype) is None:
            x = x.astype(np.float64)
        x = np.ascontiguousarray(x)
        x_dtype, n, ldx = x.dtype, x.shape[0], x.shape[1]
        x_ptr, on_host = x.ctypes.data, 1
    else:
        x_ptr = int(x)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    if conn.ndim != 2:
        raise ValueError("conn must be (n_nodes, d)")
    n_nodes, d = conn.shape
    evals = np.empty((n_nodes, d))
    evecs_cm = np.empty((n_nodes, d, d))
    mean = np.empty((n_nodes, d))
    tms = (C.c_double * 2)()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(_capi.lib().hg_sfa_train_layer(C.c_void_p(x_ptr), on_host, _capi.np_dtype_code(x_dtype), int(n), int(ldx), vp(conn), n_nodes, d,
                                               int(device), vp(evals), vp(evecs_cm), vp(mean), C.cast(tms, C.c_void_p)))
    return evals, np.swapaxes(evecs_cm, 1, 2).copy(), mean, (tms[0], tms[1])   # column-major -> [k][row, col]


def _train_layer(fn_name, x_ptr, x_dtype, n, ldx, conn, device):
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    n_nodes, d = conn.shape
    evals, evecs_cm, mean = np.empty((n_nodes, d)), np.empty((n_nodes, d, d)), np.empty((n_nodes, d))
    tms = (C.c_double * 2)()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(getattr(_capi.lib(), fn_name)(C.c_void_p(int(x_ptr)), 0, _capi.np_dtype_code(x_dtype), int(n), int(ldx), vp(conn), n_nodes, d,
                                              int(device), vp(evals), vp(evecs_cm), vp(mean), C.cast(tms, C.c_void_p)))
    return evals, np.swapaxes(evecs_cm, 1, 2).copy(), mean, (tms[0], tms[1])


def pca_train_layer(x_ptr, x_dtype, n, ldx, conn, device=0):
    """Per node: mean, eigenvalues (ascending) and orthonormal eigenvectors of Cov(x[:, conn[k]]) — device pointer in."""
    return _train_layer("hg_pca_train_layer", x_ptr, x_dtype, n, ldx, conn, device)


def train_apply(x_ptr, x_dtype, n, ldx, conn, mean, W, funcs, out_ptr, ldo, device=0):
    """out[t, node*width + f*p + j] = func_f((x[t, conn[node]] - mean[node]) @ W[node])_j in float64 on the device.
    funcs: list of (kind, exponent) with kind 0 identity / 1 |z|^e / 2 sgn(z)|z|^e, or [] for the affine map alone."""
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    W = np.ascontiguousarray(W, dtype=np.float64)
    n_nodes, d = conn.shape
    p = W.shape[2]
    kinds = np.ascontiguousarray([k for k, _ in funcs], dtype=np.int32)
    expos = np.ascontiguousarray([e for _, e in funcs], dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(_capi.lib().hg_train_apply_device(C.c_void_p(int(x_ptr)), _capi.np_dtype_code(x_dtype), int(n), int(ldx), vp(conn), n_nodes, d,
                                                  vp(mean), vp(W), p, len(funcs), vp(kinds) if len(funcs) else None,
                                                  vp(expos) if len(funcs) else None, C.c_void_p(int(out_ptr)), int(ldo), int(device)))


def train_layer_device(x, conn, p, s, funcs, device=0):
    """One layer of ``train_hierarchy_device``: per node whitening PCA (top ``p``), expansion by ``funcs``
    (element-wise ``nodes.ExpFunc``), SFA (slowest ``s``), everything on the GPU.  ``x``: (T, D) float64 torch tensor on
    the device in time order, ``conn``: (n_nodes, d_in) input columns.  Returns (mu, v, mue, sf, y): PCA means and
    whitening matrices, SFA means and matrices (host arrays, sign convention of ``synth._sign_fix``) and the layer's
    output (T, n_nodes * s) as a device tensor."""
    import torch
    from . import synth
    kind_of = {"identity": 0, "abs_pow": 1, "signed_pow": 2}
    T = x.shape[0]
    n_nodes = conn.shape[0]
    lam, vec, mu, _ = pca_train_layer(x.data_ptr(), np.float64, T, x.shape[1], conn, device)
    lam, vec = lam[:, ::-1][:, :p], vec[:, :, ::-1][:, :, :p]
    vec = synth._sign_fix(vec)
    lam = np.maximum(lam, 1e-9 * lam[:, :1])
    v = vec / np.sqrt(lam)[:, None, :]
    width = sum(f.out_dim(p) for f in funcs)
    e = torch.empty((T, n_nodes * width), dtype=torch.float64, device=x.device)
    train_apply(x.data_ptr(), np.float64, T, x.shape[1], conn, mu, v, [(kind_of[f.kind], f.expo) for f in funcs], e.data_ptr(), e.shape[1], device)
    # SFA on the expanded signal
    conn_e = np.arange(n_nodes * width, dtype=np.int32).reshape(n_nodes, width)
    _lam2, W, mue, _ = sfa_train_layer(e.data_ptr(), conn_e, device=device, x_dtype=np.float64, n=T, ldx=e.shape[1])
    sf = synth._sign_fix(W[:, :, :s])
    y = torch.empty((T, n_nodes * s), dtype=torch.float64, device=x.device)
    train_apply(e.data_ptr(), np.float64, T, e.shape[1], conn_e, mue, sf, [], y.data_ptr(), y.shape[1], device)
    return mu, v, mue, sf, y


def train_hierarchy_device(side, f0, layer_dims, n_train=1500, seed=None, expo=0.8, layout="flownode", device=0, verbose=False):
    """``synth.train_hierarchy`` (node_kind "pca_exp_sfa") with the training set resident on the GPU: per layer the node
    statistics (covariance, difference covariance: HIP kernels, fp64 accumulation), the eigen-solves (PCA on (Cov, I), SFA on
    (dCov, Cov): batched Jacobi / rocSOLVER) and the float64 pass that feeds layer l's output to layer l+1 all run on the
    device; the host only post-processes the small eigen-systems (top-p selection, sign convention, whitening scale).
    Returns the same node list as the numpy trainer (the counterpart of MDP train / stop_training)."""
    import torch
    from . import nodes as N, synth
    seed = synth.WEIGHT_SEED if seed is None else seed
    dev = torch.device("cuda", device)
    x = torch.from_numpy(synth.make_training_sequence(n_train, side, seed)).to(dev)          # (T, side*side) float64
    T = x.shape[0]
    grid = None
    flow, ch = [], 1
    for li, (p, s) in enumerate(layer_dims):
        if li == 0:
            sb = N.Rectangular2dSwitchboard((side, side), (f0, f0), (f0, f0), 1)
        else:
            nx, ny = grid
            merge_x = (li % 2 == 1)
            if merge_x and nx == 1:
                merge_x = False
            if not merge_x and ny == 1:
                merge_x = True
            field = (2, 1) if merge_x else (1, 2)
            sb = N.Rectangular2dSwitchboard((nx, ny), field, field, ch)
        grid = sb.out_channels_xy
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        conn = sb.connections.reshape(n_nodes, d_in)
        p_ = min(p, d_in)
        s_ = min(s, 2 * p_)
        funcs = [N.identity, N.unsigned_expo(expo) if expo != 0.8 else N.unsigned_08expo]
        mu, v, mue, sf, y = train_layer_device(x, conn, p_, s_, funcs, device)
        pcas = [N.WhiteningNode(mu[k], v[k]) for k in range(n_nodes)]
        exps = [N.GeneralExpansionNode(funcs, p_) for _ in range(n_nodes)]
        sfas = [N.SFANode(mue[k], sf[k]) for k in range(n_nodes)]
        if layout == "flownode":
            layer = [N.Layer([N.FlowNode([pcas[k], exps[k], sfas[k]]) for k in range(n_nodes)])]
        else:
            layer = [N.Layer(pcas), N.Layer(exps), N.Layer(sfas)]
        flow.append(sb)
        flow.extend(layer)
        x, ch = y, s_
        if verbose:
            print("  L%-2d grid %-7s nodes %4d  d_in %3d -> %3d   out std %.3f" % (li, grid, n_nodes, d_in, s_, float(x.std())))
        if n_nodes == 1 and li < len(layer_dims) - 1:
            break
    return flow
