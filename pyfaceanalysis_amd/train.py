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
