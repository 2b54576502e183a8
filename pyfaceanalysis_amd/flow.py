"""``Flow`` — the drop-in for the object the reference calls as

    sl = networks[num_network].execute(subimages_arr, benchmark=benchmark)
                                             (FaceDetectUpdated.py:699; face_analysis.py:1064,1257)

Same surface as the mdp.Flow that cuicuilco.patch_mdp extends with the ``benchmark`` kwarg:
``execute(x, nodenr=None, benchmark=None)``, ``len(flow)``, ``flow[i]``, ``input_dim`` /
``output_dim`` of the end nodes.  All arithmetic runs in the HIP library behind
include/higsfa.h; there is no CPU path here — without the library or without a GPU,
``execute`` raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .blob import blob_to_flow, flow_to_blob


class _Handle(object):
    """Owns one hg_flow*."""

    def __init__(self, blob, force_generic=False):
        L = _capi.lib()
        self._L = L
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _capi.check(L.hg_flow_load(buf, len(blob), 1 if force_generic else 0, C.byref(h)))
        self.h = h
        self.device = -1

    def info(self):
        inf = _capi.HgInfo()
        _capi.check(self._L.hg_flow_info(self.h, C.byref(inf)))
        return inf

    def describe(self):
        need = C.c_size_t()
        _capi.check(self._L.hg_flow_describe(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _capi.check(self._L.hg_flow_describe(self.h, buf, need.value, None))
        return buf.value.decode()

    def to_device(self, device):
        _capi.check(self._L.hg_flow_to_device(self.h, int(device)))
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self._L.hg_flow_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Flow(object):
    """A sequence of nodes executed on one MI355X.

    Parameters
    ----------
    flow : list of :mod:`pyfaceanalysis_amd.nodes` objects (``mdp.Flow.flow``)
    device : HIP device ordinal used at first ``execute`` (default 0)
    output_dtype : numpy dtype of the returned array; float64 mirrors MDP's node dtype
    force_generic : use the generic step-by-step plan even where the fused plan applies
    """

    def __init__(self, flow, device=0, output_dtype=np.float64, force_generic=False):
        self.flow = list(flow)
        if not self.flow:
            raise ValueError("Flow: empty node list")
        self.device = int(device)
        self.output_dtype = np.dtype(output_dtype)
        if _capi.np_dtype_code(self.output_dtype) not in (_capi.HG_F32, _capi.HG_F64):
            raise ValueError("Flow: output_dtype must be float32 or float64")
        self.force_generic = bool(force_generic)
        self._handles = {}

    # --- construction helpers -------------------------------------------------------------
    @classmethod
    def from_blob(cls, blob, **kw):
        f = cls(blob_to_flow(blob), **kw)
        f._blob_full = bytes(blob)
        return f

    def to_blob(self):
        if getattr(self, "_blob_full", None) is None:
            self._blob_full = flow_to_blob(self.flow)
        return self._blob_full

    # --- mdp.Flow container protocol --------------------------------------------------------
    def __len__(self):
        return len(self.flow)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Flow(self.flow[i], device=self.device, output_dtype=self.output_dtype,
                        force_generic=self.force_generic)
        return self.flow[i]

    def __iter__(self):
        return iter(self.flow)

    @property
    def input_dim(self):
        return self.flow[0].input_dim

    @property
    def output_dim(self):
        return self.flow[-1].output_dim

    # --- native handle management -------------------------------------------------------------
    def _handle(self, nodenr=None, on_device=True):
        key = len(self.flow) - 1 if nodenr is None else int(nodenr)
        if not 0 <= key < len(self.flow):
            raise ValueError("nodenr %r out of range for a flow of %d nodes" % (nodenr, len(self.flow)))
        h = self._handles.get(key)
        if h is None:
            blob = self.to_blob() if key == len(self.flow) - 1 else flow_to_blob(self.flow[:key + 1])
            h = _Handle(blob, self.force_generic)
            self._handles[key] = h
        if on_device and h.device < 0:
            h.to_device(self.device)
        return h

    def info(self, nodenr=None):
        return self._handle(nodenr).info()

    def describe(self, nodenr=None):
        """Plan listing (role of more_nodes.describe_flow, FaceDetectUpdated.py:193)."""
        return self._handle(nodenr).describe()

    def host_transport(self, nodenr=None):
        """How the last ``execute`` on host rows reached the device: 1 = packer threads stored straight into device memory (large
        BAR and every input buffer confirmed host-mapped by hsa_amd_pointer_info), 0 = pinned ring + copy queues, -1 = no host call
        yet (hg_flow_host_transport)."""
        t = C.c_int(-1)
        _capi.check(_capi.lib().hg_flow_host_transport(self._handle(nodenr).h, C.byref(t)))
        return t.value

    def host_plan(self):
        """Parse + plan on the host only (works without a GPU): returns hg_info."""
        h = _Handle(self.to_blob(), self.force_generic)
        try:
            return h.info(), h.describe()
        finally:
            h.close()

    # --- the hot call -----------------------------------------------------------------------------
    def execute(self, x, nodenr=None, benchmark=None, n_cols=None, devices=None):
        """Process ``x`` (N, input_dim) through the nodes up to ``nodenr`` (all by default).

        ``benchmark``: object with the reference's Benchmark interface (benchmarking.py:39-58);
        when given and enabled, per-stage GPU times are recorded with
        ``add_task_ellapsed(label, seconds, reference)`` like cuicuilco's patched ``_execute_seq``.
        ``n_cols`` (extension): return only the first n_cols features — the caller consumes
        ``sl[:, 0:classifier.input_dim]`` (FaceDetectUpdated.py:709,719).
        ``devices`` (extension): list of HIP device ordinals; the rows are cut into len(devices) contiguous
        blocks that run concurrently, one replica of the weights per entry (``hg_flow_execute_sharded``).
        """
        x = np.asarray(x)
        if x.ndim != 2:
            raise _capi.NodeException("x has rank %d, should be 2" % x.ndim)
        if x.shape[1] != self.input_dim:
            raise _capi.NodeException("x has dimension %d, should be %d" % (x.shape[1], self.input_dim))
        if nodenr is not None and not 0 <= int(nodenr) < len(self.flow):
            raise ValueError("nodenr %r out of range for a flow of %d nodes" % (nodenr, len(self.flow)))
        out_dim = self.flow[len(self.flow) - 1 if nodenr is None else nodenr].output_dim
        h = self._handle(nodenr, on_device=devices is None)
        code = _capi.np_dtype_code(x.dtype)
        if code is None:
            x = x.astype(np.float64)
            code = _capi.HG_F64
        if x.strides[1] != x.dtype.itemsize or (x.shape[0] > 1 and x.strides[0] % x.dtype.itemsize) \
                or x.strides[0] < x.shape[1] * x.dtype.itemsize:
            x = np.ascontiguousarray(x)           # F-ordered / sliced inputs: one host copy
        ldx = x.strides[0] // x.dtype.itemsize if x.shape[0] > 1 else x.shape[1]
        cols = out_dim if n_cols is None else int(n_cols)
        if not 0 < cols <= out_dim:
            raise ValueError("n_cols must be in 1..%d" % out_dim)
        n = x.shape[0]
        y = np.empty((n, cols), dtype=self.output_dtype)
        if n == 0:
            return y
        L = _capi.lib()
        prof = benchmark is not None and getattr(benchmark, "enabled", True)
        if prof:
            _capi.check(L.hg_flow_reset_profile(h.h))
        _capi.check(L.hg_flow_set_profiling(h.h, 1 if prof else 0))
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _capi.check(L.hg_flow_execute_sharded(h.h, x.ctypes.data_as(C.c_void_p), code, n, ldx, y.ctypes.data_as(C.c_void_p),
                                                  _capi.np_dtype_code(y.dtype), cols, cols, devs, len(devices)))
            return y
        _capi.check(L.hg_flow_execute(h.h, x.ctypes.data_as(C.c_void_p), code, n, ldx,
                                      y.ctypes.data_as(C.c_void_p), _capi.np_dtype_code(y.dtype), cols, cols))
        if prof:
            for name, ms, _cnt in self.stage_times(nodenr):
                benchmark.add_task_ellapsed(name, ms * 1e-3, getattr(benchmark, "default_reference", None))
        return y

    __call__ = execute

    def stage_times(self, nodenr=None):
        """[(stage name, total ms, launches)] accumulated since the last profiled execute began."""
        h = self._handle(nodenr)
        L = _capi.lib()
        ns = C.c_int()
        _capi.check(L.hg_flow_stage_times(h.h, None, None, 0, C.byref(ns)))
        ms = (C.c_double * ns.value)()
        cnt = (C.c_int64 * ns.value)()
        _capi.check(L.hg_flow_stage_times(h.h, ms, cnt, ns.value, C.byref(ns)))
        out = []
        for i in range(ns.value):
            buf = C.create_string_buffer(512)      # (stage names carry the plan's notes: up to ~350 characters)
            _capi.check(L.hg_flow_stage_name(h.h, i, buf, 512))
            out.append((buf.value.decode(), ms[i], cnt[i]))
        return out

    # --- device-resident entry (benchmarks, chained pipelines) ------------------------------------
    def reserve(self, max_rows, nodenr=None):
        h = self._handle(nodenr)
        _capi.check(_capi.lib().hg_flow_reserve(h.h, int(max_rows)))

    def execute_device(self, x_ptr, x_dtype, n, ldx, y_ptr, y_dtype, y_cols, ldy, stream=0, nodenr=None,
                       profile=False):
        """Enqueue on ``stream`` with raw device pointers (ints); no synchronisation."""
        h = self._handle(nodenr)
        L = _capi.lib()
        _capi.check(L.hg_flow_set_profiling(h.h, 1 if profile else 0))
        _capi.check(L.hg_flow_execute_device(
            h.h, C.c_void_p(x_ptr), _capi.np_dtype_code(x_dtype), int(n), int(ldx), C.c_void_p(y_ptr),
            _capi.np_dtype_code(y_dtype), int(y_cols), int(ldy), C.c_void_p(stream)))

    def close(self):
        for h in self._handles.values():
            h.close()
        self._handles = {}
