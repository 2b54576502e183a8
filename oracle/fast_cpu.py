"""TEST ORACLE — NOT PRODUCT CODE.  *** parity unpinned *** (see mdp_restate.py header).

Driver of oracle/fast_cpu.c, the "good CPU" point of bench.py's cpu_baseline leg:
flattens a flow (Switchboard / Layer / FlowNode / PCA / SFA / GeneralExpansion with element-wise
functions) into one op list and runs it in C over row chunks with OpenMP.  Anything else raises.
tests/test_oracle.py requires it to agree with mdp_restate to 1e-12.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Op(C.Structure):
    _fields_ = [("kind", C.c_long), ("src", C.c_long), ("src_off", C.c_long), ("dst", C.c_long),
                ("dst_off", C.c_long), ("d_in", C.c_long), ("d_out", C.c_long), ("p0", C.c_long),
                ("expo", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfast_cpu.so")
        src = os.path.join(_HERE, "fast_cpu.c")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                                os.path.getmtime(src[:-2] + "_pow.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(so)
        L.fc_run.argtypes = [C.POINTER(_Op), C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_long),
                             C.POINTER(C.c_long), C.c_long, C.c_long, C.POINTER(C.c_double), C.c_long,
                             C.POINTER(C.c_double), C.c_long, C.c_int]
        L.fc_run.restype = C.c_int
        _LIB = L
    return _LIB


def _names(obj):
    return [c.__name__ for c in type(obj).__mro__]


class Plan:
    """Flat op list.  Buffers 0/1 ping-pong between top-level nodes; 2/3 are per-node scratch."""

    def __init__(self, flow_nodes):
        self.ops, self.dpool, self.lpool = [], [], []
        self.width = [int(flow_nodes[0].input_dim), 1, 1, 1]
        cur = 0
        for node in flow_nodes:
            nxt = 1 - cur
            self.width[nxt] = max(self.width[nxt], int(node.output_dim))
            self._emit(node, cur, 0, nxt, 0, depth=0)
            cur = nxt
        self.out_buf, self.out_dim = cur, int(flow_nodes[-1].output_dim)
        self._ops = (_Op * len(self.ops))(*self.ops)
        self._d = np.ascontiguousarray(np.concatenate(self.dpool) if self.dpool else np.zeros(1))
        self._l = np.ascontiguousarray(np.concatenate(self.lpool) if self.lpool else np.zeros(1, np.int64), np.int64)
        self._w = np.asarray(self.width, dtype=np.int64)

    def _dp(self, *arrs):
        off = sum(a.size for a in self.dpool)
        for a in arrs:
            self.dpool.append(np.asarray(a, dtype=np.float64).reshape(-1))
        return off

    def _affine(self, a, W, b, src, so, dst, do):
        W = np.asarray(W, dtype=np.float64)
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        b = np.asarray(b, dtype=np.float64).reshape(-1)
        p0 = self._dp(np.broadcast_to(a, (W.shape[0],)), W, np.broadcast_to(b, (W.shape[1],)))
        self.ops.append(_Op(1, src, so, dst, do, W.shape[0], W.shape[1], p0, 0.0))

    def _emit(self, node, src, so, dst, do, depth):
        names = _names(node)
        if "Switchboard" in names:
            idx = np.asarray(node.connections, dtype=np.int64)
            off = sum(a.size for a in self.lpool)
            self.lpool.append(idx)
            self.ops.append(_Op(0, src, so, dst, do, int(node.input_dim), idx.size, off, 0.0))
        elif "Layer" in names:
            i0 = o0 = 0
            for sub in node.nodes:
                self._emit(sub, src, so + i0, dst, do + o0, depth)
                i0 += int(sub.input_dim)
                o0 += int(sub.output_dim)
        elif "FlowNode" in names:
            if depth:
                raise TypeError("fast_cpu: nested FlowNode")
            subs = list(node.flow)
            s, o = src, so
            for k, sub in enumerate(subs):
                if k == len(subs) - 1:
                    d, dd = dst, do
                else:
                    d, dd = (2 if s != 2 else 3), 0
                    self.width[d] = max(self.width[d], int(sub.output_dim))
                self._emit(sub, s, o, d, dd, depth + 1)
                s, o = d, dd
        elif "PCANode" in names:
            self._affine(node.avg, node.v, 0.0, src, so, dst, do)
        elif "SFANode" in names:
            self._affine(0.0, node.sf, -np.asarray(node._bias).reshape(-1), src, so, dst, do)
        elif "GeneralExpansionNode" in names:
            o = 0
            d = int(node.input_dim)
            for f in node.funcs:
                kind = {"identity": 2, "abs_pow": 3, "signed_pow": 4}.get(f.kind)
                if kind is None or (f.sel > 0 and f.sel < d):
                    raise TypeError("fast_cpu: expansion %r not covered" % (f.kind,))
                self.ops.append(_Op(kind, src, so, dst, do + o, d, d, 0, float(f.expo)))
                o += d
        else:
            raise TypeError("fast_cpu: no plan for %s" % type(node).__name__)

    def run(self, x, threads=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape[1] != self.width[0]:
            raise ValueError("fast_cpu: dimension mismatch")
        y = np.empty((x.shape[0], self.out_dim))
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_long)
        rc = lib().fc_run(self._ops, len(self.ops), self._d.ctypes.data_as(dp), self._l.ctypes.data_as(lp),
                          self._w.ctypes.data_as(lp), len(self.width), self.out_buf, x.ctypes.data_as(dp),
                          x.shape[0], y.ctypes.data_as(dp), self.out_dim, int(threads or os.cpu_count() or 1))
        if rc != 0:
            raise MemoryError("fast_cpu: scratch allocation failed")
        return y


def execute_flow(flow_nodes, x, threads=None):
    return Plan(flow_nodes).run(x, threads)
