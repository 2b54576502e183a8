"""TEST ORACLE — NOT PRODUCT CODE.

numpy restatement of the sub-image extraction rule of SURVEY.md §8f-1, the producer of the hot call's input:

    subimages = extract_subimages_rotate(images, idx, coords, -1 * curr_angles, (w, h), Image.NEAREST)
    subimages_arr = images_asarray(subimages)                    (face_analysis.py:781-786, FaceDetectUpdated.py:686)

``cuicuilco.image_loader.extract_subimages_rotate`` is not in /root/reference.  What IS available
and pinned is the third-party library it drives, PIL: ``tests/test_extract_rule.py`` checks every function here
bit for bit against ``PIL.Image.rotate`` / ``PIL.Image.transform`` themselves, so this file is a restatement of
PIL's arithmetic (Pillow src/libImaging/Geometry.c: ImagingScaleAffine, affine_fixed; PIL/Image.py: Image.rotate),
not of cuicuilco's.  The composition rule the build owns ([K], documented in DESIGN.md §f1):

    window = frame.rotate(delta_ang, NEAREST, center = centre of the box)          when delta_ang != 0
             .transform((w, h), EXTENT, (x0, y0, x1, y1), NEAREST)

with delta_ang = -curr_angle as the reference passes it (face_analysis.py:782), counter-clockwise positive (PIL).
"""
from __future__ import annotations

import math

import numpy as np


def extent_table(lo, hi, m, lim):
    """Source index of every output pixel along one axis for Image.transform(EXTENT, NEAREST)
    (ImagingScaleAffine): a = (hi - lo) / m; o = lo + a / 2; then o += a per pixel — the running SUM, not
    lo + a (i + 0.5) — and index = -1 if o < 0 else int(o); indices outside [0, lim) read as 'no pixel' (-1)."""
    a = (hi - lo) / m
    o = lo + a * 0.5
    out = np.empty(m, dtype=np.int64)
    for i in range(m):
        v = -1 if o < 0.0 else int(o)
        out[i] = v if 0 <= v < lim else -1
        o += a
    return out


def _fix(v):
    """Geometry.c FIX(): 16.16 fixed point, FLOOR(v * 65536 + 0.5) with FLOOR(x) = (int)x for x >= 0 else floor(x)."""
    x = v * 65536.0 + 0.5
    return int(x) if x >= 0.0 else int(math.floor(x))


def rotate_coeffs(angle, cx, cy):
    """Image.rotate(angle, center=(cx, cy)) -> the six 16.16 fixed-point coefficients affine_fixed() steps with,
    or None when PIL takes its 'scaling' branch instead (sin rounds to 0: multiples of 180 degrees).
    Python side (Image.rotate): matrix from cos / sin of -radians(angle % 360) rounded to 15 decimals."""
    angle = angle % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    if m[1] == 0 and m[3] == 0:
        return None, m
    A0, A1, A3, A4 = _fix(m[0]), _fix(m[1]), _fix(m[3]), _fix(m[4])
    A2 = _fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    A5 = _fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return (A0, A1, A2, A3, A4, A5), m


def rotated_source(xr, yr, coeffs, m, fw, fh):
    """Pixel (xr, yr) of frame.rotate(...) -> (xs, ys) in the frame, or (-1, -1) for fill.
    affine_fixed: xx = A2 + yr A1 + xr A0 (integer sums), xs = xx >> 16, same for y.
    Scaling branch (ImagingScaleAffine with a[0] = +-1): xo = a2 + a0 / 2 then += a0 per pixel."""
    if coeffs is not None:
        A0, A1, A2, A3, A4, A5 = coeffs
        xs = (A2 + yr * A1 + xr * A0) >> 16
        ys = (A5 + yr * A4 + xr * A3) >> 16
    else:
        xo = m[2] + m[0] * 0.5
        for _ in range(xr):
            xo += m[0]
        yo = m[5] + m[4] * 0.5
        for _ in range(yr):
            yo += m[4]
        xs = -1 if xo < 0.0 else int(xo)
        ys = -1 if yo < 0.0 else int(yo)
    if 0 <= xs < fw and 0 <= ys < fh:
        return xs, ys
    return -1, -1


def extract_subimages_rotate(frame, boxes, delta_angs, out_size):
    """(N, w*h) matrix of windows (row-major pixels, images_asarray layout), dtype of ``frame``."""
    frame = np.asarray(frame)
    fh, fw = frame.shape
    w, h = out_size
    out = np.zeros((len(boxes), w * h), dtype=frame.dtype)
    for n, (b, ang) in enumerate(zip(boxes, delta_angs)):
        tx = extent_table(b[0], b[2], w, fw)
        ty = extent_table(b[1], b[3], h, fh)
        if ang % 360.0 == 0.0:          # no rotation (cuicuilco skips the rotate step; PIL's would be the identity)
            for y in range(h):
                if ty[y] < 0:
                    continue
                ok = tx >= 0
                out[n, y * w:(y + 1) * w][ok] = frame[ty[y], tx[ok]]
            continue
        coeffs, m = rotate_coeffs(ang, (b[0] + b[2]) / 2.0, (b[1] + b[3]) / 2.0)
        for y in range(h):
            if ty[y] < 0:
                continue
            for x in range(w):
                if tx[x] < 0:
                    continue
                xs, ys = rotated_source(int(tx[x]), int(ty[y]), coeffs, m, fw, fh)
                if xs >= 0:
                    out[n, y * w + x] = frame[ys, xs]
    return out
