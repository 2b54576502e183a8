"""Gaussian-classifier soft-label regression — the step right after the hot call:

    reg_out = classifiers[num_network].regression(sl[:, 0:reg_num_signals], avg_labels)
                                                               (FaceDetectUpdated.py:709-719)

``GaussianClassifier`` carries the attributes of the mdp.nodes.GaussianClassifier objects stored
in ``SavedClassifiers/*.pckl`` (``means``, ``inv_covs``, ``_sqrt_def_covs``, ``p``, ``labels``,
``avg_labels``, ``input_dim``) and evaluates ``regression`` on the GPU through the C ABI
(``hg_gauss_*``, include/higsfa.h).  ``load_classifier_pickle`` reads the reference's Python-2
pickles with a stub unpickler (no mdp import).
"""
from __future__ import annotations

import ctypes as C
import pickle

import numpy as np

from . import _capi


class GaussianClassifier(object):
    def __init__(self, means, inv_covs, sqrt_def_covs, p, labels=None, avg_labels=None, device=0):
        self.means = np.ascontiguousarray(means, dtype=np.float64)
        self.inv_covs = np.ascontiguousarray(inv_covs, dtype=np.float64)
        self._sqrt_def_covs = np.ascontiguousarray(sqrt_def_covs, dtype=np.float64).reshape(-1)
        self.p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        k, d = self.means.shape
        if self.inv_covs.shape != (k, d, d) or self._sqrt_def_covs.shape != (k,) or self.p.shape != (k,):
            raise ValueError("GaussianClassifier: inconsistent parameter shapes")
        self.labels = np.arange(k) if labels is None else np.asarray(labels)
        self.avg_labels = None if avg_labels is None else np.ascontiguousarray(avg_labels, dtype=np.float64).reshape(-1)
        self.input_dim = d
        self.device = int(device)
        self._h = None
        self._h_avg = None

    def _handle(self, avg_labels):
        avg = np.ascontiguousarray(avg_labels, dtype=np.float64).reshape(-1)
        if avg.shape != self.p.shape:
            raise ValueError("avg_labels must have one entry per class")
        if self._h is not None and np.array_equal(avg, self._h_avg):
            return self._h
        self.close()
        L = _capi.lib()
        h = C.c_void_p()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _capi.check(L.hg_gauss_create(self.means.shape[0], self.input_dim, vp(self.means), vp(self.inv_covs),
                                      vp(self._sqrt_def_covs), vp(self.p), vp(avg), self.device, C.byref(h)))
        self._h, self._h_avg = h, avg.copy()
        return h

    def regression(self, x, avg_labels=None, estimate_std=False):
        """Soft-label regression ``sum_c P(c|x) avg_labels[c]`` (and its posterior std)."""
        if avg_labels is None:
            avg_labels = self.avg_labels
        if avg_labels is None:
            raise ValueError("regression needs avg_labels")
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != self.input_dim:
            raise _capi.NodeException("x has shape %r, classifier input_dim is %d" % (x.shape, self.input_dim))
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        x = np.ascontiguousarray(x)
        n = x.shape[0]
        reg = np.empty(n)
        sd = np.empty(n) if estimate_std else None
        if n:
            h = self._handle(avg_labels)
            _capi.check(_capi.lib().hg_gauss_regression(
                h, x.ctypes.data_as(C.c_void_p), _capi.np_dtype_code(x.dtype), n, x.shape[1],
                reg.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p) if estimate_std else None))
        return (reg, sd) if estimate_std else reg

    def regression_device(self, x_ptr, x_dtype, n, ldx, out_reg_ptr, out_std_ptr=None, stream=0, avg_labels=None):
        """Device-resident form (raw pointers, enqueued on ``stream``): reads the first ``input_dim`` columns of the
        (n, ldx) feature matrix — the caller's ``sl[:, 0:reg_num_signals]`` (FaceDetectUpdated.py:719) — and writes n
        float64 regression outputs."""
        h = self._handle(self.avg_labels if avg_labels is None else avg_labels)
        _capi.check(_capi.lib().hg_gauss_regression_device(
            h, C.c_void_p(x_ptr), _capi.np_dtype_code(x_dtype), int(n), int(ldx), C.c_void_p(out_reg_ptr),
            C.c_void_p(out_std_ptr) if out_std_ptr else None, C.c_void_p(stream)))

    def close(self):
        if self._h is not None:
            _capi.lib().hg_gauss_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Stub(object):
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"_state": st})


# The ONLY globals the stub reader resolves to real objects: what numpy arrays / scalars and plain containers
# need to be rebuilt.  Everything else — mdp.*, cuicuilco modules, the aliases of FaceDetectUpdated.py:57-68,
# but also builtins such as eval / getattr / __import__ and any other numpy helper — becomes an inert
# attribute bag, so reading a third-party .pckl executes no code of its choosing.
_SAFE_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("copyreg", "_reconstructor"), ("_codecs", "encode"), ("collections", "OrderedDict"),
}
_SAFE_GLOBALS |= {("builtins", n) for n in ("object", "list", "dict", "tuple", "set", "frozenset", "int", "float", "complex",
                                             "str", "bytes", "bytearray", "slice", "bool")}
_PY2_MODULES = {"__builtin__": "builtins", "copy_reg": "copyreg"}
_PY2_NAMES = {("builtins", "long"): "int", ("builtins", "unicode"): "str"}


class StubUnpickler(pickle.Unpickler):
    """Resolves the allowlisted numpy / builtin globals normally and every other global to an attribute-bag
    class carrying its module and name."""

    def find_class(self, module, name):
        mod = _PY2_MODULES.get(module, module)
        nm = _PY2_NAMES.get((mod, name), name)
        if (mod, nm) in _SAFE_GLOBALS:
            return super(StubUnpickler, self).find_class(mod, nm)
        return type(str(name), (_Stub,), {"__module__": module})


def load_stub_pickle(path):
    with open(path, "rb") as f:
        obj = StubUnpickler(f, encoding="latin1").load()
    if isinstance(obj, tuple):      # cache files may hold (object, ...) — face_analysis.py:473-478
        obj = obj[0]
    return obj


def classifier_from_stub(obj, device=0):
    if type(obj).__name__ != "GaussianClassifier":
        raise TypeError("expected a pickled GaussianClassifier, found %s.%s" % (type(obj).__module__, type(obj).__name__))
    d = obj.__dict__
    return GaussianClassifier(d["means"], d["inv_covs"], d["_sqrt_def_covs"], d["p"], d.get("labels"),
                              d.get("avg_labels"), device=device)


def load_classifier_pickle(path, device=0):
    return classifier_from_stub(load_stub_pickle(path), device=device)
