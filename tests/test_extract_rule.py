"""CPU: the extraction rule restated in oracle/pil_restate.py against PIL itself (Image.rotate + Image.transform,
the library cuicuilco.image_loader.extract_subimages_rotate drives — face_analysis.py:781-786), bit for bit."""
import numpy as np
import pytest
from PIL import Image

from oracle import pil_restate


def pil_windows(frame, boxes, angs, size):
    im = Image.fromarray(frame, "L")
    out = []
    for b, a in zip(boxes, angs):
        src = im
        if a % 360.0 != 0.0:
            src = im.rotate(a, Image.NEAREST, center=((b[0] + b[2]) / 2.0, (b[1] + b[3]) / 2.0))
        out.append(np.asarray(src.transform(size, Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1))
    return np.stack(out)


def test_rule_matches_pil():
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    boxes, angs = [], []
    for _ in range(60):
        x0, y0 = rng.uniform(-10, 100), rng.uniform(-10, 70)
        s = rng.uniform(8, 60)
        boxes.append([x0, y0, x0 + s - 1, y0 + s * rng.uniform(0.8, 1.2) - 1])
        angs.append(float(rng.choice([0.0, 5.0, -7.3, 22.5, -22.5, 90.0, 180.0, 270.0, 360.0, 45.0, 13.37, -180.0, 181.0, 1e-9])))
    boxes = np.asarray(boxes)
    for size in ((16, 16), (21, 13)):
        ref = pil_windows(frame, boxes, angs, size)
        got = pil_restate.extract_subimages_rotate(frame, boxes, angs, size)
        bad = np.nonzero((ref != got).any(axis=1))[0]
        assert bad.size == 0, (size, [(int(i), angs[i]) for i in bad[:5]])


def test_rotation_really_rotates():
    """A window rotated by 90 degrees about its centre is the transposed-and-flipped window (sanity of the convention:
    PIL angles are counter-clockwise)."""
    frame = np.arange(40 * 40, dtype=np.uint8).reshape(40, 40)
    box = np.array([[10.0, 10.0, 30.0, 30.0]])
    w0 = pil_restate.extract_subimages_rotate(frame, box, [0.0], (20, 20)).reshape(20, 20)
    w90 = pil_restate.extract_subimages_rotate(frame, box, [90.0], (20, 20)).reshape(20, 20)
    assert not np.array_equal(w0, w90)
    assert np.array_equal(w90, pil_windows(frame, box, [90.0], (20, 20)).reshape(20, 20))
