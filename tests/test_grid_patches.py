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
    for size in ((64, 64), (128, 128), (37, 21)):
        boxes = np.concatenate([b for _, b in grid.frame_boxes(1000, 562, 0.2, subimage_size=(64, 64))])[::3]
        extra = np.array([[-5.5, -3.25, 40.0, 30.0], [950.0, 520.0, 1020.5, 580.0], [10.0, 10.0, 11.0, 11.0]])   # partly outside
        boxes = np.vstack([boxes, extra])
        ref = np.stack([np.asarray(im.transform(size, Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1) for b in boxes])
        for dt in (np.uint8, np.float32, np.float64):
            got = p.extract(frame, boxes, size, dtype=dt)
            assert got.dtype == dt and got.shape == (len(boxes), size[0] * size[1])
            assert np.array_equal(got.astype(np.int64), ref.astype(np.int64))
    assert p.extract(frame, np.zeros((0, 4)), (64, 64)).shape == (0, 4096)
    with pytest.raises(NotImplementedError):
        p.extract(frame, boxes[:2], (64, 64), angles=[0.0, 5.0])
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
