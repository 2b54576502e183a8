"""The detection cascade around the hot call, resident on one GPU (BASELINE.json configs[2]).

The reference runs, per image, per pyramid level and per cascade stage (FaceDetectUpdated.py:592-766):

    subimages_arr = load_network_subimages(...)                      :686   extract (PIL, host)
    sl = networks[k].execute(subimages_arr, benchmark=benchmark)     :699   THE HOT CALL
    reg_out = classifiers[k].regression(sl[:, 0:d], avg_labels)      :719
    update_current_subimage_coordinates / identify_patches_to_discard :728-735 (face_analysis.py:803-887)
    boolean-mask compaction of every per-candidate array              :739-759

``DeviceCascade`` runs the same loop inside the library (``hg_cascade_detect_device``, include/higsfa.h: rotated window
extraction -> ``hg_flow_execute_device`` -> Gaussian regression -> one fused update / discard / compaction kernel per stage)
on one stream: no per-candidate array ever visits the host; the host reads one integer after each Disc stage (the survivor
count, which shrinks a lot there and sizes the next launches).  All pyramid levels run as ONE batch of candidates, as the
reference's author notes is possible (:599), each candidate carrying its level's constants.  torch is used for the frame
tensor and the stream only.  (The single steps are also exported one by one — ``hg_cascade_update_device``,
``hg_cascade_compact_device``, ``hg_gather_rows_device`` — for callers that drive the loop themselves.)

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
        self._h = None
        for st in self.stages:
            if st.classifier.input_dim > self.k:
                raise ValueError("stage %s: classifier reads %d features, cascade keeps %d" % (st.name, st.classifier.input_dim, self.k))

    def _consts(self):
        p = self.pipeline
        c = _capi.HgCascadeConsts()
        c.regression_width, c.regression_height = p["regression_width"], p["regression_height"]
        c.desired_sampling = DESIRED_SAMPLING
        c.tolerance_posxy_deviation, c.tolerance_scale_deviation, c.tolerance_angle_deviation = TOLERANCE_POSXY, TOLERANCE_SCALE, TOLERANCE_ANGLE
        c.max_scale_radio, c.min_scale_radio = p["net_maxs"] / 0.825, p["net_mins"] / 0.825
        c.net_Dang = p["net_Dang"]
        c.cut_off_face = 0.0                 # per stage: cut_offs_face[serial]
        return c

    def _handle(self):
        if self._h is not None:
            return self._h
        L = _capi.lib()
        arr = (_capi.HgCascadeStage * len(self.stages))()
        for i, st in enumerate(self.stages):
            arr[i].type, arr[i].serial = _capi.HG_STAGE[st.type], st.serial
            if st.flow is not None:
                if st.flow.device != self.device or st.flow.output_dtype != np.float32:
                    raise ValueError("stage %s: flow must live on device %d with output_dtype float32" % (st.name, self.device))
                arr[i].flow = st.flow._handle().h
            else:
                arr[i].flow = None
            arr[i].classifier = st.classifier._handle(st.classifier.avg_labels)
        cut = (C.c_double * len(CUT_OFFS_FACE))(*CUT_OFFS_FACE)
        cc = self._consts()
        h = C.c_void_p()
        _capi.check(L.hg_cascade_create(arr, len(self.stages), self.w, self.h, self.k, C.byref(cc), cut, len(CUT_OFFS_FACE), self.device, C.byref(h)))
        self._h = h
        return h

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
        orig_index (n), confidence (n), counts (survivors after every stage, -1 where the count stayed on the device),
        rows_executed."""
        t, L = self.torch, _capi.lib()
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        if frame.dtype != t.uint8 or frame.device != self.dev or frame.stride(1) != 1:
            raise ValueError("frame must be a uint8 tensor on %s with contiguous rows" % (self.dev,))
        boxes, level = frame_windows(fw, fh, smallest_face, self.pipeline, (self.w, self.h)) if windows is None else windows
        boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        level = np.ascontiguousarray(level, dtype=np.float64)
        n0 = len(boxes)
        coords, angles = np.empty((n0, 4)), np.empty(n0)
        oidx, conf = np.empty(n0, dtype=np.int32), np.empty(n0)
        counts = np.zeros(len(self.stages), dtype=np.int32)
        n_out, rows = C.c_int64(), C.c_int64()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _capi.check(L.hg_cascade_detect_device(
            self._handle(), C.c_void_p(frame.data_ptr()), fh, fw, frame.stride(0), vp(boxes), vp(level), n0, vp(coords), vp(angles), vp(oidx),
            vp(conf), n0, C.byref(n_out), vp(counts), C.byref(rows), C.c_void_p(t.cuda.current_stream(self.dev).cuda_stream)))
        n = n_out.value
        return dict(coords=coords[:n].copy(), angles=angles[:n].copy(), orig_index=oidx[:n].astype(np.int64), confidence=conf[:n].copy(),
                    counts=[int(c) for c in counts], rows_executed=int(rows.value), n_windows=n0)

    def close(self):
        if self._h is not None:
            _capi.lib().hg_cascade_free(self._h)
            self._h = None
        self.patcher.close()
