"""Grid builders (CPU, known answers from SURVEY.md §6) and on-device patch extraction (GPU, bit-exact vs PIL)."""
import numpy as np
import pytest

from pyfaceanalysis_amd import grid


def _counts(w, h, smallest_face):
    levels = grid.frame_boxes(w, h, smallest_face)
    return len(levels), sum(len(b) for _, b in levels), max(len(b) for _, b in levels)


def test_grid_known_answers():
    """SURVEY.md §6, computed from the reference's own formulas (face_analysis.py:575-669) with the
    constants of Pipeline_experimental.txt:2: a 1920x1080 frame gives 10 levels / 1738 windows / largest
    batch 728 when prescaled to 1000x562, 1729 without prescaling; smallest_face 0.2 gives 7 / 386 / 169."""
    assert grid.prescaled_size(1920, 1080) == (1000, 562)
    assert _counts(1000, 562, 0.1) == (10, 1738, 728)
    assert _counts(1920, 1080, 0.1)[:2] == (10, 1729)
    assert _counts(1000, 562, 0.2) == (7, 386, 169)


def test_grid_boxes_shape_and_order():
    s, boxes = grid.frame_boxes(1000, 562, 0.2)[0]
    pw = 64 * s
    assert np.allclose(boxes[:, 2] - boxes[:, 0], pw - 1) and np.allclose(boxes[:, 3] - boxes[:, 1], pw - 1)
    assert boxes[0, 0] == 0.0 and boxes[0, 1] == 0.0 and np.isclose(boxes[-1, 2], 1000 - 1) and np.isclose(boxes[-1, 3], 562 - 1)
    n_x = np.unique(boxes[:, 0]).size
    assert np.all(np.diff(boxes[:n_x, 0]) > 0) and np.all(boxes[:n_x, 1] == boxes[0, 1])      # x runs fastest (y-major)


@pytest.mark.gpu
def test_patch_extraction_matches_pil(native_lib):
    from PIL import Image
    from pyfaceanalysis_amd.patches import Patcher
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, (562, 1000), dtype=np.uint8)
    im = Image.fromarray(frame, "L")
    p = Patcher()
    # (uint8 windows whose rows are a multiple of 16 pixels take the sixteen-pixels-per-thread kernel — 8-byte loads where four source
    # pixels lie within eight bytes, byte loads where they do not: windows narrower and wider than their output, at the frame's right
    # edge, partly outside; (48, 5), (37, 21): the four-pixels-per-thread kernel)
    for size in ((64, 64), (128, 128), (37, 21), (16, 16), (256, 32), (128, 100), (48, 5)):
        boxes = np.concatenate([b for _, b in grid.frame_boxes(1000, 562, 0.2, subimage_size=(64, 64))])[::3]
        extra = np.array([[-5.5, -3.25, 40.0, 30.0], [950.0, 520.0, 1020.5, 580.0], [10.0, 10.0, 11.0, 11.0],   # partly outside / tiny
                          [0.0, 0.0, 999.0, 561.0], [960.0, 500.0, 999.0, 561.0], [990.0, 3.0, 999.9, 9.0], [100.25, 50.5, 163.75, 120.0],
                          [3.0, 3.0, 950.0, 40.0]])
        boxes = np.vstack([boxes, extra])
        ref = np.stack([np.asarray(im.transform(size, Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1) for b in boxes])
        for dt in (np.uint8, np.float32, np.float64):
            got = p.extract(frame, boxes, size, dtype=dt)
            assert got.dtype == dt and got.shape == (len(boxes), size[0] * size[1])
            assert np.array_equal(got.astype(np.int64), ref.astype(np.int64))
    assert p.extract(frame, np.zeros((0, 4)), (64, 64)).shape == (0, 4096)
    # a frame's first stage: all 1738 windows of the 1080p grid at 128 x 128 (whole windows per workgroup from 1024 boxes on; fewer boxes
    # above took row chunks), and a handful of boxes (one row pass per workgroup)
    boxes = np.concatenate([b for _, b in grid.frame_boxes(1000, 562, 0.1, subimage_size=(128, 128))])
    assert len(boxes) == 1738
    got = p.extract(frame, boxes, (128, 128), dtype=np.uint8)
    for i in list(range(0, 1738, 13)) + [1737]:
        assert np.array_equal(got[i], np.asarray(im.transform((128, 128), Image.EXTENT, tuple(boxes[i]), Image.NEAREST)).reshape(-1)), i
    few = p.extract(frame, boxes[[0, 900, 1737]], (128, 128), dtype=np.uint8)
    assert np.array_equal(few, got[[0, 900, 1737]])
    p.close()


def _pil_rotated_windows(frame, boxes, angs, size):
    """The rule hg_extract.hip implements, evaluated by PIL itself: rotate about the box centre, then EXTENT."""
    from PIL import Image
    im = Image.fromarray(frame, "L")
    out = []
    for b, a in zip(boxes, angs):
        src = im if a % 360.0 == 0.0 else im.rotate(a, Image.NEAREST, center=((b[0] + b[2]) / 2.0, (b[1] + b[3]) / 2.0))
        out.append(np.asarray(src.transform(size, Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1))
    return np.stack(out)


@pytest.mark.gpu
def test_rotated_patch_extraction_matches_pil(native_lib):
    """Windows with delta_ang != 0 (every stage after the first PAng, face_analysis.py:781-783): bit-exact against
    PIL's Image.rotate + Image.transform for the angle range of the cascade (|angle| <= net_Dang * 1.1 = 24.75 per
    iteration, three iterations), right angles, multiples of 180 (PIL's scaling branch), tiny angles, boxes partly
    outside the frame; and against the numpy restatement in oracle/pil_restate.py."""
    from oracle import pil_restate
    from pyfaceanalysis_amd.patches import Patcher
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, (562, 1000), dtype=np.uint8)
    boxes = np.concatenate([b for _, b in grid.frame_boxes(1000, 562, 0.2, subimage_size=(64, 64))])[::2]
    extra = np.array([[-5.5, -3.25, 40.0, 30.0], [950.0, 520.0, 1020.5, 580.0], [10.0, 10.0, 11.0, 11.0], [300.0, 200.0, 363.0, 263.0]])
    boxes = np.vstack([boxes, extra])
    n = len(boxes)
    angs = rng.uniform(-75.0, 75.0, n)
    special = [0.0, 90.0, -90.0, 180.0, -180.0, 270.0, 360.0, 540.0, 1e-9, -1e-9, 22.5, -22.5, 45.0, 0.5]
    angs[:len(special)] = special
    angs[-4:] = [33.0, -12.0, 180.0, 181.0]
    p = Patcher()
    # ((128, 128): the pipelines' size; (16, 7), (32, 10): rows of sixteen pixels whose table rows are not all 16-byte aligned — the sixteen
    # table entries then come as scalar loads; (37, 21): the four-pixels-per-thread path)
    for size in ((64, 64), (37, 21), (128, 128), (16, 7), (32, 10)):
        ref = _pil_rotated_windows(frame, boxes, angs, size)
        got = p.extract(frame, boxes, size, dtype=np.uint8, delta_angs=angs)
        bad = np.nonzero((got != ref).any(axis=1))[0]
        assert bad.size == 0, (size, [(int(i), float(angs[i]), int((got[i] != ref[i]).sum())) for i in bad[:8]])
    sub = slice(0, 40)
    assert np.array_equal(pil_restate.extract_subimages_rotate(frame, boxes[sub], angs[sub], (64, 64)),
                          p.extract(frame, boxes[sub], (64, 64), dtype=np.uint8, delta_angs=angs[sub]))
    # float frames / float64 windows, and an all-zero angle vector equals the plain call
    gf = p.extract(frame.astype(np.float32), boxes[:50], (64, 64), dtype=np.float64, delta_angs=angs[:50])
    assert np.array_equal(gf, _pil_rotated_windows(frame, boxes[:50], angs[:50], (64, 64)).astype(np.float64))
    assert np.array_equal(p.extract(frame, boxes, (64, 64), dtype=np.uint8, delta_angs=np.zeros(n)), p.extract(frame, boxes, (64, 64), dtype=np.uint8))
    with pytest.raises(ValueError):
        p.extract(frame, boxes[:2], (64, 64), delta_angs=[0.0])
    p.close()


@pytest.mark.gpu
def test_frame_to_features_pipeline(native_lib, nets):
    """Frame -> windows -> flow features, all through the HIP library; checked against PIL + oracle."""
    from PIL import Image
    from oracle import mdp_restate as oracle
    from pyfaceanalysis_amd.flow import Flow
    from pyfaceanalysis_amd.patches import Patcher
    nodes = nets("T5L-16")
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (120, 200), dtype=np.uint8)
    boxes = grid.level_boxes(200, 120, 0.5, 64, 64, 128, 128, 40.0, 20.0)
    x = Patcher().extract(frame, boxes, (16, 16), dtype=np.uint8)
    im = Image.fromarray(frame, "L")
    xr = np.stack([np.asarray(im.transform((16, 16), Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1) for b in boxes])
    assert np.array_equal(x, xr)
    ref = oracle.execute_flow(nodes, xr)
    y = Flow(nodes).execute(x)
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()
