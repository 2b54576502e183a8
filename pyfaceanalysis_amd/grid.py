"""Sliding-window grid of the detection cascade: which sub-images reach the hot call.

Restates (vectorised, adaptive non-tracking branch only) the three grid builders of the reference,
which decide N for every ``flow.execute`` call (SURVEY.md §2.1 "grid builders", Appendix B):

* ``compute_sampling_values``                         face_analysis.py:575-607
* ``compute_posX_posY_values``                        face_analysis.py:610-657
* ``compute_subimage_coordinates_from_posX_posY_values``   face_analysis.py:661-669

and the constants of ``Pipelines/Pipeline_experimental.txt:2`` / FaceDetectUpdated.py:84,110-111,121-122.
These are O(N) host formulas; they are here so that a frame can be turned into the batches of
BASELINE.json config 3 without the reference's script.  The tracking branches
(``track_single_face``) and the non-adaptive branches are out of scope.
"""
from __future__ import annotations

import math

import numpy as np

# Pipelines/Pipeline_experimental.txt:2  (net_Dx net_Dy net_Dang net_mins net_maxs subW subH regW regH)
FACE_PIPELINE = dict(net_Dx=40.0, net_Dy=20.0, net_Dang=22.5, net_mins=0.694, net_maxs=0.981,
                     subimage_width=64, subimage_height=64, regression_width=128, regression_height=128)
PATCH_OVERLAP_SAMPLING = 1.1        # FaceDetectUpdated.py:110
PATCH_OVERLAP_POSX_POSY = 1.1       # FaceDetectUpdated.py:111
PRESCALE_SIZE = 1000                # FaceDetectUpdated.py:122


def prescaled_size(width, height, prescale_size=PRESCALE_SIZE):
    """FaceDetectUpdated.py:551-556: shrink so that the larger side is <= prescale_size."""
    f = max(width * 1.0 / prescale_size, height * 1.0 / prescale_size)
    if f > 1.0:
        return int(width / f), int(height / f)
    return width, height


def sampling_values(im_width, im_height, subimage_width, subimage_height, smallest_face, net_mins, net_maxs,
                    patch_overlap_sampling=PATCH_OVERLAP_SAMPLING):
    """Pyramid levels (face_analysis.py:586-598): start at the smallest face box (>= 20 px), grow by
    (net_maxs / net_mins) / overlap while the largest face of the level still fits the image."""
    min_box_side = max(20, min(im_height, im_width) * smallest_face * 0.825 / net_mins)
    s = min_box_side * 1.0 / subimage_width
    step = (net_maxs / net_mins) / patch_overlap_sampling
    out = []
    while subimage_width * s * net_mins / 0.825 < im_width and subimage_height * s * net_mins / 0.825 < im_height:
        out.append(s)
        s *= step
    return out


def level_boxes(im_width, im_height, sampling_value, subimage_width, subimage_height, regression_width, regression_height,
                net_Dx, net_Dy, patch_overlap_posx_posy=PATCH_OVERLAP_POSX_POSY):
    """(N, 4) boxes (x0, y0, x1, y1) of one pyramid level, y-major like the reference
    (face_analysis.py:630-646 grid, :661-669 box = (posX, posY, posX + pw - 1, posY + ph - 1))."""
    pw, ph = subimage_width * sampling_value, subimage_height * sampling_value
    sep_x = net_Dx * 2.0 * pw / regression_width
    sep_y = net_Dy * 2.0 * ph / regression_height
    nx = int(math.ceil((1 + (im_width - pw) / sep_x) * patch_overlap_posx_posy))
    ny = int(math.ceil((1 + (im_height - ph) / sep_y) * patch_overlap_posx_posy))
    xs = np.linspace(0.0, im_width - pw, nx)
    ys = np.linspace(0.0, im_height - ph, ny)
    gx, gy = np.meshgrid(xs, ys)            # row (y) major
    x0, y0 = gx.reshape(-1), gy.reshape(-1)
    return np.stack([x0, y0, x0 + pw - 1, y0 + ph - 1], axis=1)


def frame_boxes(im_width, im_height, smallest_face=0.2, pipeline=None, subimage_size=None):
    """All first-stage windows of one frame: list of (sampling_value, boxes) per pyramid level.
    ``subimage_size`` overrides the pipeline's 64x64 sub-image (BASELINE.json measures 128x128)."""
    p = dict(FACE_PIPELINE if pipeline is None else pipeline)
    if subimage_size is not None:
        p["subimage_width"], p["subimage_height"] = subimage_size
    levels = []
    for s in sampling_values(im_width, im_height, p["subimage_width"], p["subimage_height"], smallest_face,
                             p["net_mins"], p["net_maxs"]):
        levels.append((s, level_boxes(im_width, im_height, s, p["subimage_width"], p["subimage_height"],
                                      p["regression_width"], p["regression_height"], p["net_Dx"], p["net_Dy"])))
    return levels
