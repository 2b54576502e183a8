"""The detection cascade around the hot call, resident on one GPU (BASELINE.json configs[2]).

The reference runs, per image, per pyramid level and per cascade stage (FaceDetectUpdated.py:592-766):

    subimages_arr = load_network_subimages(...)                      :686   extract (PIL, host)
    sl = networks[k].execute(subimages_arr, benchmark=benchmark)     :699   THE HOT CALL
    reg_out = classifiers[k].regression(sl[:, 0:d], avg_labels)      :719
    update_current_subimage_coordinates / identify_patches_to_discard :728-735 (face_analysis.py:803-887)
    boolean-mask compaction of every per-candidate array              :739-759

``DeviceCascade`` chains the same steps through the C ABI (``hg_patcher_extract_rotate_device`` ->
``hg_flow_execute_device`` -> ``hg_gauss_regression_device`` -> ``hg_cascade_update_device`` ->
``hg_cascade_compact_device`` / ``hg_gather_rows_device``) on one stream: no per-candidate array ever visits the
host; the host reads one integer per stage (the survivor count, needed to size the next launches).  All pyramid
levels run as ONE batch of candidates, as the reference's author notes is possible (:599), each candidate carrying its
level's constants.  torch is used for device buffers and the stream only.

Stages follow the pipeline grammar (face_analysis.py:437-443): a type with a serial digit ("Disc1", "PosX0", ...), a flow
or None (the stage reuses the previous features, :680-682), a classifier.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi, grid
from .patches import Patcher

CUT_OFFS_FACE = [0.99, 0.95, 0.85, 0.8, 0.7, 0.6, 0.5, 0.45, 0.10, 0.05]      # FaceDetectUpdated.py:98
TOLERANCE_SCALE = TOLERANCE_ANGLE = TOLERANCE_POSXY = 1.1                     # FaceDetectUpdated.py:113-115
DESIRED_SAMPLING = 0.825                                                      # FaceDetectUpdated.py:729


class Stage(object):
    def __init__(self, name, flow, classifier):
        self.name = name
        self.type, self.serial = name[:-1], int(name[-1])
        if self.type not in _capi.HG_STAGE:
            raise ValueError("unknown stage type %r" % name)
        self.flow, self.classifier = flow, classifier


def frame_windows(im_width, im_height, smallest_face, pipeline, subimage_size):
    """All first-stage windows of a frame (grid.frame_boxes) with, per window, the constants of its pyramid level:
    (max_Dx_diff, max_Dy_diff, base_side) — face_analysis.py:651-652, FaceDetectUpdated.py:604-605."""
    p = dict(pipeline)
    p["subimage_width"], p["subimage_height"] = subimage_size
    boxes, level = [], []
    for s, b in grid.frame_boxes(im_width, im_height, smallest_face, pipeline=p):
        pw, ph = p["subimage_width"] * s, p["subimage_height"] * s
        boxes.append(b)
        level.append(np.tile([p["net_Dx"] * pw / p["regression_width"], p["net_Dy"] * ph / p["regression_height"],
                              math.sqrt(pw ** 2 + ph ** 2)], (len(b), 1)))
    return np.concatenate(boxes), np.concatenate(level)


class DeviceCascade(object):
    def __init__(self, stages, subimage_size, n_features, pipeline=None, device=0):
        import torch
        self.torch = torch
        self.stages = list(stages)
        self.w, self.h = int(subimage_size[0]), int(subimage_size[1])
        self.k = int(n_features)              # feature columns kept per candidate (>= every classifier's input_dim)
        self.pipeline = dict(grid.FACE_PIPELINE if pipeline is None else pipeline)
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        self.patcher = Patcher(self.device)
        self.cap = 0
        for st in self.stages:
            if st.classifier.input_dim > self.k:
                raise ValueError("stage %s: classifier reads %d features, cascade keeps %d" % (st.name, st.classifier.input_dim, self.k))
        self._pinned_count = torch.zeros(1, dtype=torch.int32).pin_memory()

    def _consts(self, serial):
        p = self.pipeline
        c = _capi.HgCascadeConsts()
        c.regression_width, c.regression_height = p["regression_width"], p["regression_height"]
        c.desired_sampling = DESIRED_SAMPLING
        c.tolerance_posxy_deviation, c.tolerance_scale_deviation, c.tolerance_angle_deviation = TOLERANCE_POSXY, TOLERANCE_SCALE, TOLERANCE_ANGLE
        c.max_scale_radio, c.min_scale_radio = p["net_maxs"] / 0.825, p["net_mins"] / 0.825
        c.net_Dang = p["net_Dang"]
        c.cut_off_face = CUT_OFFS_FACE[serial]
        return c

    def _reserve(self, n0):
        if n0 <= self.cap:
            return
        t, d = self.torch, self.dev
        pp = lambda *shape, dtype: [t.empty(shape, dtype=dtype, device=d) for _ in range(2)]     # ping-pong pairs
        self.coords, self.angles = pp(n0, 4, dtype=t.float64), pp(n0, dtype=t.float64)
        self.oidx, self.conf = pp(n0, dtype=t.int32), pp(n0, dtype=t.float64)
        self.sl, self.subs = pp(n0, self.k, dtype=t.float32), pp(n0, self.w * self.h, dtype=t.uint8)
        self.reg = t.empty(n0, dtype=t.float64, device=d)
        self.discard = t.empty(n0, dtype=t.uint8, device=d)
        self.map = t.empty(n0, dtype=t.int32, device=d)
        self.count = t.zeros(1, dtype=t.int32, device=d)
        self.orig_coords = t.empty((n0, 4), dtype=t.float64, device=d)
        self.orig_level = t.empty((n0, 3), dtype=t.float64, device=d)
        self.orig_angles = t.zeros(n0, dtype=t.float64, device=d)
        self.neg = t.empty(n0, dtype=t.float64, device=d)
        self.cap = n0
        for st in self.stages:
            if st.flow is not None:
                st.flow.reserve(n0)

    def prescale(self, frame, prescale_size=grid.PRESCALE_SIZE):
        """FaceDetectUpdated.py:551-556: shrink so that the larger side is <= prescale_size, ``Image.resize(NEAREST)`` —
        PIL's nearest resize is the EXTENT rule over the whole frame, so the patcher does it (bit-exact vs PIL in the tests)."""
        t = self.torch
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        pw, ph = grid.prescaled_size(fw, fh, prescale_size)
        if (pw, ph) == (fw, fh):
            return frame
        whole = t.tensor([[0.0, 0.0, float(fw), float(fh)]], dtype=t.float64, device=self.dev)
        small = t.empty((ph, pw), dtype=t.uint8, device=self.dev)
        self.patcher.extract_device(frame.data_ptr(), np.uint8, fh, fw, frame.stride(0), whole.data_ptr(), 1, (pw, ph), small.data_ptr(),
                                    np.uint8, pw * ph, stream=t.cuda.current_stream(self.dev).cuda_stream)
        self._keep = whole          # alive until the kernel has run
        return small

    def detect(self, frame, smallest_face=0.2, windows=None):
        """frame: (H, W) uint8 torch tensor on this device.  Returns a dict of host arrays: coords (n, 4), angles (n),
        orig_index (n), confidence (n), counts (survivors after every stage), rows_executed."""
        t, L = self.torch, _capi.lib()
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        if frame.dtype != t.uint8 or frame.device != self.dev or frame.stride(1) != 1:
            raise ValueError("frame must be a uint8 tensor on %s with contiguous rows" % (self.dev,))
        boxes, level = frame_windows(fw, fh, smallest_face, self.pipeline, (self.w, self.h)) if windows is None else windows
        n = n0 = len(boxes)
        self._reserve(n0)
        stream = t.cuda.current_stream(self.dev)
        sp = stream.cuda_stream
        self.orig_coords[:n0].copy_(t.from_numpy(np.ascontiguousarray(boxes)), non_blocking=True)
        self.orig_level[:n0].copy_(t.from_numpy(np.ascontiguousarray(level)), non_blocking=True)
        cur = 0
        self.coords[cur][:n0].copy_(self.orig_coords[:n0])
        self.angles[cur][:n0].zero_()
        self.oidx[cur][:n0].copy_(t.arange(n0, dtype=t.int32, device=self.dev))
        self.conf[cur][:n0].zero_()
        counts, rows_executed = [], 0
        vp = lambda x: C.c_void_p(x.data_ptr())
        for k, st in enumerate(self.stages):
            if n == 0:
                counts.append(0)
                continue
            skip_extract = (k > 0 and self.stages[k - 1].type == "Disc") or st.flow is None          # FaceDetectUpdated.py:674-681
            if not skip_extract:
                t.neg(self.angles[cur][:n], out=self.neg[:n])                                         # -1 * curr_angles (face_analysis.py:782)
                self.patcher.extract_device(frame.data_ptr(), np.uint8, fh, fw, frame.stride(0), self.coords[cur].data_ptr(), n,
                                            (self.w, self.h), self.subs[cur].data_ptr(), np.uint8, self.w * self.h, stream=sp,
                                            delta_angs_ptr=self.neg.data_ptr())
            if st.flow is not None:
                st.flow.execute_device(self.subs[cur].data_ptr(), np.uint8, n, self.w * self.h, self.sl[cur].data_ptr(), np.float32,
                                       self.k, self.k, stream=sp)
                rows_executed += n
            st.classifier.regression_device(self.sl[cur].data_ptr(), np.float32, n, self.k, self.reg.data_ptr(), stream=sp)
            cc = self._consts(st.serial)
            _capi.check(L.hg_cascade_update_device(self.device, _capi.HG_STAGE[st.type], C.byref(cc), n, vp(self.coords[cur]), vp(self.angles[cur]),
                                                   vp(self.reg), vp(self.oidx[cur]), vp(self.orig_coords), vp(self.orig_angles),
                                                   vp(self.orig_level), vp(self.discard), C.c_void_p(sp)))
            _capi.check(L.hg_cascade_compact_device(self.device, vp(self.discard), n, vp(self.map), vp(self.count), C.c_void_p(sp)))
            nxt = 1 - cur
            rows = [(self.coords, 32), (self.angles, 8), (self.oidx, 4), (self.sl, 4 * self.k)]
            # the sub-images are reused only by a stage that follows a Disc stage and has a flow of its own (:674-677)
            if st.type == "Disc" and k + 1 < len(self.stages) and self.stages[k + 1].flow is not None:
                rows.append((self.subs, self.w * self.h))
            for buf, rb in rows:
                _capi.check(L.hg_gather_rows_device(self.device, vp(buf[cur]), vp(buf[nxt]), rb, vp(self.map), vp(self.count), n, C.c_void_p(sp)))
            src_conf = self.reg if st.type == "Disc" else self.conf[cur]                               # :758-759
            _capi.check(L.hg_gather_rows_device(self.device, vp(src_conf), vp(self.conf[nxt]), 8, vp(self.map), vp(self.count), n, C.c_void_p(sp)))
            cur = nxt
            self._pinned_count.copy_(self.count, non_blocking=True)        # the one host readback of the stage
            stream.synchronize()
            n = int(self._pinned_count[0])
            counts.append(n)
        out = dict(coords=self.coords[cur][:n].cpu().numpy(), angles=self.angles[cur][:n].cpu().numpy(),
                   orig_index=self.oidx[cur][:n].cpu().numpy().astype(np.int64), confidence=self.conf[cur][:n].cpu().numpy(),
                   counts=counts, rows_executed=rows_executed, n_windows=n0)
        return out

    def close(self):
        self.patcher.close()
