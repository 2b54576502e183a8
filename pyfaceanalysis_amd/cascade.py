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


def frame_levels(im_width, im_height, smallest_face, pipeline, subimage_size):
    """The same grid as a table of pyramid levels (``_capi.HgCascadeLevel``) for ``hg_cascade_detect_levels_device``, which
    computes the windows on the device: per level the grid counts and linspace end points of face_analysis.py:640-646, the patch
    size (:630-631) and the level constants (:651-652).  ``frame_windows`` of the same arguments is what the device produces,
    bit for bit (tests/test_cascade.py)."""
    p = dict(pipeline)
    p["subimage_width"], p["subimage_height"] = subimage_size
    svals = grid.sampling_values(im_width, im_height, p["subimage_width"], p["subimage_height"], smallest_face, p["net_mins"], p["net_maxs"])
    arr = (_capi.HgCascadeLevel * max(len(svals), 1))()
    for i, s in enumerate(svals):
        pw, ph = p["subimage_width"] * s, p["subimage_height"] * s
        sep_x = p["net_Dx"] * 2.0 * pw / p["regression_width"]
        sep_y = p["net_Dy"] * 2.0 * ph / p["regression_height"]
        arr[i].nx = int(math.ceil((1 + (im_width - pw) / sep_x) * grid.PATCH_OVERLAP_POSX_POSY))
        arr[i].ny = int(math.ceil((1 + (im_height - ph) / sep_y) * grid.PATCH_OVERLAP_POSX_POSY))
        arr[i].x_stop, arr[i].y_stop = im_width - pw, im_height - ph
        arr[i].patch_w, arr[i].patch_h = pw, ph
        arr[i].max_dx = p["net_Dx"] * pw / p["regression_width"]
        arr[i].max_dy = p["net_Dy"] * ph / p["regression_height"]
        arr[i].base_side = math.sqrt(pw ** 2 + ph ** 2)
    return arr, len(svals), sum(arr[i].nx * arr[i].ny for i in range(len(svals)))


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
        self._levels = {}        # (frame size, smallest_face) -> level table (frame_levels)
        self._prescale = {}      # frame size -> (box tensor, output tensor) of the prescale step
        self._frames = {}        # (frame size, smallest_face, prescale size) -> everything detect_frame needs per call
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
        if (fw, fh) not in self._prescale:      # the box of the whole frame and the output live with the cascade: no per-frame upload / allocation
            self._prescale[(fw, fh)] = (t.tensor([[0.0, 0.0, float(fw), float(fh)]], dtype=t.float64, device=self.dev),
                                        t.empty((ph, pw), dtype=t.uint8, device=self.dev))
        whole, small = self._prescale[(fw, fh)]
        self.patcher.extract_device(frame.data_ptr(), np.uint8, fh, fw, frame.stride(0), whole.data_ptr(), 1, (pw, ph), small.data_ptr(),
                                    np.uint8, pw * ph, stream=t.cuda.current_stream(self.dev).cuda_stream)
        return small

    def detect(self, frame, smallest_face=0.2, windows=None):
        """frame: (H, W) uint8 torch tensor on this device.  Returns a dict of host arrays: coords (n, 4), angles (n),
        orig_index (n), confidence (n), counts (survivors after every stage, -1 where the count stayed on the device),
        rows_executed."""
        t, L = self.torch, _capi.lib()
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        if frame.dtype != t.uint8 or frame.device != self.dev or frame.stride(1) != 1:
            raise ValueError("frame must be a uint8 tensor on %s with contiguous rows" % (self.dev,))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        counts = np.zeros(len(self.stages), dtype=np.int32)
        n_out, rows = C.c_int64(), C.c_int64()
        stream = C.c_void_p(t.cuda.current_stream(self.dev).cuda_stream)
        if windows is None:
            # the frame's own grid: the windows are computed on the device from the table of pyramid levels (face_analysis.py:630-669)
            key = (fw, fh, float(smallest_face))
            if key not in self._levels:
                self._levels[key] = frame_levels(fw, fh, smallest_face, self.pipeline, (self.w, self.h))
            levels, n_levels, n0 = self._levels[key]
            coords, angles = np.empty((n0, 4)), np.empty(n0)
            oidx, conf = np.empty(n0, dtype=np.int32), np.empty(n0)
            _capi.check(L.hg_cascade_detect_levels_device(
                self._handle(), C.c_void_p(frame.data_ptr()), fh, fw, frame.stride(0), levels, n_levels, vp(coords), vp(angles), vp(oidx),
                vp(conf), n0, C.byref(n_out), vp(counts), C.byref(rows), stream))
        else:
            boxes, level = windows
            boxes = np.ascontiguousarray(boxes, dtype=np.float64)
            level = np.ascontiguousarray(level, dtype=np.float64)
            n0 = len(boxes)
            coords, angles = np.empty((n0, 4)), np.empty(n0)
            oidx, conf = np.empty(n0, dtype=np.int32), np.empty(n0)
            _capi.check(L.hg_cascade_detect_device(
                self._handle(), C.c_void_p(frame.data_ptr()), fh, fw, frame.stride(0), vp(boxes), vp(level), n0, vp(coords), vp(angles), vp(oidx),
                vp(conf), n0, C.byref(n_out), vp(counts), C.byref(rows), stream))
        n = n_out.value
        return dict(coords=coords[:n].copy(), angles=angles[:n].copy(), orig_index=oidx[:n].astype(np.int64), confidence=conf[:n].copy(),
                    counts=[int(c) for c in counts], rows_executed=int(rows.value), n_windows=n0)

    def detect_frame(self, frame, smallest_face=0.2, prescale_size=grid.PRESCALE_SIZE):
        """prescale + detect as ONE host call (hg_cascade_detect_frame_device): what the reference does per image between loading
        it and writing its detections (FaceDetectUpdated.py:551-561 prescale, :589-600 grid, :665-766 stage loop).  Everything that
        depends only on the frame size — prescaled size, level table, output buffers — is computed once and kept."""
        t, L = self.torch, _capi.lib()
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        key = (fw, fh, float(smallest_face), int(prescale_size or 0))
        plan = self._frames.get(key)
        if plan is None:
            if frame.dtype != t.uint8 or frame.device != self.dev or frame.stride(1) != 1:
                raise ValueError("frame must be a uint8 tensor on %s with contiguous rows" % (self.dev,))
            pw, ph = grid.prescaled_size(fw, fh, prescale_size) if prescale_size else (fw, fh)
            pre = (pw, ph) if (pw, ph) != (fw, fh) else (0, 0)
            levels, n_levels, n0 = frame_levels(pw, ph, smallest_face, self.pipeline, (self.w, self.h))
            bufs = (np.empty((n0, 4)), np.empty(n0), np.empty(n0, dtype=np.int32), np.empty(n0), np.zeros(len(self.stages), dtype=np.int32))
            plan = self._frames[key] = (pre, levels, n_levels, n0, bufs, [b.ctypes.data_as(C.c_void_p) for b in bufs], C.c_int64(), C.c_int64())
        pre, levels, n_levels, n0, (coords, angles, oidx, conf, counts), ptr, n_out, rows = plan
        _capi.check(L.hg_cascade_detect_frame_device(
            self._handle(), frame.data_ptr(), fh, fw, frame.stride(0), pre[0], pre[1], levels, n_levels, ptr[0], ptr[1], ptr[2], ptr[3], n0,
            C.byref(n_out), ptr[4], C.byref(rows), t.cuda.current_stream(self.dev).cuda_stream))
        n = n_out.value
        return dict(coords=coords[:n].copy(), angles=angles[:n].copy(), orig_index=oidx[:n].astype(np.int64), confidence=conf[:n].copy(),
                    counts=counts.tolist(), rows_executed=rows.value, n_windows=n0)

    def close(self):
        if self._h is not None:
            _capi.lib().hg_cascade_free(self._h)
            self._h = None
        self.patcher.close()
