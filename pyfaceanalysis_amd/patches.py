"""Sub-image extraction on the GPU: ``Patcher.extract`` is the counterpart of

    subimages = extract_subimages_rotate(images, idx, coords, -angles, (w, h), Image.NEAREST)
    subimages_arr = images_asarray(subimages)                       (face_analysis.py:781-786)

same pixels as ``PIL.Image.transform((w, h), EXTENT, box, NEAREST)`` — of the frame itself, or for a window with
a non-zero ``delta_ang`` of ``frame.rotate(delta_ang, NEAREST, center=box centre)`` (the composition rule is the
build's: hg_extract.hip header) — returned as the (N, w*h) row-major matrix ``flow.execute`` takes.  No CPU path:
needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


class Patcher(object):
    def __init__(self, device=0):
        self.device = int(device)
        self._h = None

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            _capi.check(_capi.lib().hg_patcher_create(self.device, C.byref(h)))
            self._h = h
        return self._h

    def extract(self, frame, boxes, out_size, dtype=np.float64, delta_angs=None):
        """frame: (H, W) uint8 or float32; boxes: (N, 4) (x0, y0, x1, y1); out_size: (w, h); delta_angs: (N,) degrees,
        what the reference passes as ``-1 * curr_angles`` (face_analysis.py:782)."""
        frame = np.asarray(frame)
        if frame.ndim != 2:
            raise ValueError("frame must be a 2-d (H, W) array")
        if frame.dtype not in (np.uint8, np.float32):
            frame = frame.astype(np.float32)
        frame = np.ascontiguousarray(frame)
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 4)
        w, h = int(out_size[0]), int(out_size[1])
        n = boxes.shape[0]
        angs = None
        if delta_angs is not None:
            angs = np.ascontiguousarray(delta_angs, dtype=np.float64).reshape(-1)
            if angs.shape[0] != n:
                raise ValueError("delta_angs must have one entry per box")
        out = np.empty((n, w * h), dtype=dtype)
        code = _capi.np_dtype_code(out.dtype)
        if code is None:
            raise ValueError("dtype must be uint8, float32 or float64")
        if n:
            _capi.check(_capi.lib().hg_patcher_extract_rotate(
                self._handle(), frame.ctypes.data_as(C.c_void_p), _capi.np_dtype_code(frame.dtype), frame.shape[0], frame.shape[1],
                frame.shape[1], boxes.ctypes.data_as(C.c_void_p), None if angs is None else angs.ctypes.data_as(C.c_void_p), n, w, h,
                out.ctypes.data_as(C.c_void_p), code, w * h))
        return out

    def extract_device(self, frame_ptr, frame_dtype, frame_h, frame_w, ld, boxes_ptr, n, out_size, out_ptr, out_dtype, ldo, stream=0,
                       delta_angs_ptr=None):
        """Raw device pointers (ints); enqueued on ``stream``, no synchronisation."""
        w, h = int(out_size[0]), int(out_size[1])
        _capi.check(_capi.lib().hg_patcher_extract_rotate_device(
            self._handle(), C.c_void_p(frame_ptr), _capi.np_dtype_code(frame_dtype), int(frame_h), int(frame_w), int(ld),
            C.c_void_p(boxes_ptr), C.c_void_p(delta_angs_ptr) if delta_angs_ptr else None, int(n), w, h, C.c_void_p(out_ptr),
            _capi.np_dtype_code(out_dtype), int(ldo), C.c_void_p(stream)))

    def close(self):
        if self._h is not None:
            _capi.lib().hg_patcher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
