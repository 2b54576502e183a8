"""The detection cascade around the hot call (BASELINE.json configs[2]): cascade glue restated from the reference
(CPU, known answers) and the device-resident chain against it (GPU)."""
import numpy as np
import pytest

from oracle import cascade_restate as CR
from pyfaceanalysis_amd import grid


def test_glue_known_answers():
    """update_current_subimage_coordinates / identify_patches_to_discard (face_analysis.py:803-887) on hand-computed cases."""
    c = np.array([[10.0, 20.0, 73.0, 83.0], [100.0, 50.0, 227.0, 177.0]])
    a = np.array([0.0, 3.0])
    cx, _ = CR.update_coordinates("PosX", c.copy(), a.copy(), np.array([12.8, -6.4]), 128, 128)
    assert np.allclose(cx[:, [0, 2]], [[10 - 12.8 * 63 / 128, 73 - 12.8 * 63 / 128], [100 + 6.4 * 127 / 128, 227 + 6.4 * 127 / 128]])
    assert np.array_equal(cx[:, [1, 3]], c[:, [1, 3]])
    _, an = CR.update_coordinates("PAng", c.copy(), a.copy(), np.array([5.0, -1.0]), 128, 128)
    assert np.array_equal(an, [5.0, 2.0])
    cs, _ = CR.update_coordinates("Scale", c.copy(), a.copy(), np.array([0.825, 0.4125]), 128, 128)
    assert np.allclose(cs[0], c[0]) and np.allclose(cs[1], [163.5 - 127, 113.5 - 127, 163.5 + 127, 113.5 + 127])
    cd, ad = CR.update_coordinates("Disc", c.copy(), a.copy(), np.array([0.1, 0.9]), 128, 128)
    assert np.array_equal(cd, c) and np.array_equal(ad, a)
    oi = np.array([0, 1])
    lvl = np.array([[20.0, 10.0, 89.1], [40.0, 20.0, 179.6]])
    w = CR.patches_to_discard("PosX", cx, a, None, oi, c, np.zeros(2), lvl, 0.694, 0.981, 22.5, 0.5)
    assert list(w) == [False, False]
    far = cx.copy()
    far[0, [0, 2]] += 28.4                  # centre now 22.1 px from the original one: limit 20 * 1.1
    assert list(CR.patches_to_discard("PosX", far, a, None, oi, c, np.zeros(2), lvl, 0.694, 0.981, 22.5, 0.5)) == [True, False]
    assert list(CR.patches_to_discard("PAng", c, np.array([24.74, -24.76]), None, oi, c, np.zeros(2), lvl, 0.694, 0.981, 22.5, 0.5)) == [False, True]
    assert list(CR.patches_to_discard("Disc", c, a, np.array([0.94, 0.95]), oi, c, np.zeros(2), lvl, 0.694, 0.981, 22.5, 0.95)) == [False, True]
    big = c.copy()
    big[1] = [0.0, 0.0, 300.0, 300.0]
    assert list(CR.patches_to_discard("Scale", big, a, None, oi, c, np.zeros(2), lvl, 0.694, 0.981, 22.5, 0.5)) == [False, True]


def test_stage_loop_on_cpu_callables():
    """The stage loop (FaceDetectUpdated.py:665-766) with toy callables: extraction only where the reference extracts,
    networks only where the pipeline has one, survivors compacted after every stage."""
    names = ["Disc1", "PosX0", "PosY0", "Disc3", "Disc9"]
    has_net = [True, True, False, True, True]
    boxes = np.array([[i * 10.0, 0.0, i * 10.0 + 31, 31.0] for i in range(6)])
    level = np.tile([10.0, 5.0, 45.25], (6, 1))
    calls = {"extract": 0, "execute": []}

    def extract(coords, dang):
        calls["extract"] += 1
        return coords[:, :1] * np.ones((1, 4))

    def execute(k, subs):
        calls["execute"].append((k, len(subs)))
        return subs[:, :2] / 50.0

    def regress(k, sl):
        return sl[:, 0]                       # Disc1: cut-off 0.95 -> windows 0..4 stay (x0/50 < 0.95)
    out = CR.run_cascade(names, has_net, boxes, level, grid.FACE_PIPELINE, extract, execute, regress)
    # extraction: Disc1, Disc3 (previous is PosY); NOT PosX0 / Disc9 (previous is Disc), NOT PosY0 (no network)
    assert calls["extract"] == 2 and [k for k, _ in calls["execute"]] == [0, 1, 3, 4]
    assert out["counts"][0] == 5 and out["counts"][-1] <= out["counts"][0]
    assert len(out["coords"]) == out["counts"][-1] == len(out["confidence"])


# stage sequences the shipped pipeline does not contain but hg_cascade_create accepts: sub-images extracted at one stage and
# reused after a Disc stage WITHOUT a network, or after two Disc stages in a row of which the first had no reason to pass
# them on (the reference compacts subimages_arr at every stage, FaceDetectUpdated.py:753, and reuses it, :674-681)
AWKWARD = {
    "pipeline": None,
    "disc_without_network_then_reuse": [("Disc1", True, 9), ("PosX0", True, 10), ("Disc3", False, 9), ("PosX1", True, 10), ("PosY1", False, 10),
                                        ("Disc5", True, 9), ("Disc7", False, 9), ("Disc9", True, 9)],
    "two_disc_then_reuse": [("Disc1", True, 9), ("Disc3", False, 9), ("Disc5", False, 9), ("PosX0", True, 10), ("PAng0", False, 10),
                            ("Scale0", True, 10), ("Disc7", False, 9), ("PosY0", False, 10), ("Disc9", True, 9), ("PosX1", True, 10)],
    # round 5 (stage groups: a network stage and the None stages behind it run as one regression launch + one glue launch, at most
    # four stages): seven stages on ONE sl — the second group's first stage is a None stage NOT preceded by a network stage of its
    # own group — with Disc stages inside both groups (their survivor counts must still reach the host)
    "long_run_on_one_sl": [("Disc1", True, 9), ("PosX0", True, 10), ("PosY0", False, 10), ("Disc3", False, 9), ("PAng0", False, 10),
                           ("Scale0", False, 10), ("PosX1", False, 10), ("Disc5", False, 9), ("Disc7", True, 9), ("PosY1", False, 10), ("Disc9", True, 9)],
}


@pytest.mark.gpu
@pytest.mark.parametrize("sequence", sorted(AWKWARD))
def test_device_cascade_matches_restated_loop(native_lib, nets, sequence):
    """DeviceCascade (extract -> execute -> regression -> update -> discard -> compaction, all on the GPU) against the
    restated stage loop driven by PIL windows and the SAME device features (so only the glue is compared): identical
    survivor sets, counts, coordinates and angles; the device regression of EVERY stage against the C restatement
    (oracle/ref_c.c gauss_regression) on the same features; then the features themselves against the oracle.
    Four distinct flow handles play the pipeline's four flows (synth_cascade.FLOW_ROLE)."""
    import torch
    from oracle import mdp_restate, pil_restate
    from pyfaceanalysis_amd import synth_cascade
    from pyfaceanalysis_amd.cascade import DeviceCascade, frame_windows
    from pyfaceanalysis_amd.flow import Flow
    from pyfaceanalysis_amd.patches import Patcher
    from oracle import ref_c
    nodes = nets("T5L-16")
    flow = Flow(nodes, output_dtype=np.float32)
    nodes4 = [nodes] + [nets("T5L-16", seed=777 + i) for i in range(3)]
    flows = [flow] + [Flow(nd, output_dtype=np.float32) for nd in nodes4[1:]]
    rng = np.random.default_rng(21)
    frame = np.rint(rng.integers(0, 256, (90, 160)).astype(np.float64)).astype(np.uint8)
    pipe = dict(grid.FACE_PIPELINE)
    K = 10                                   # T5L-16 has 10 outputs: classifier widths capped to it
    boxes, level = frame_windows(160, 90, 0.3, pipe, (16, 16))
    pt = Patcher()
    subs0 = pt.extract(frame, boxes, (16, 16), dtype=np.uint8)
    feats = flow.execute(subs0)
    stages = synth_cascade.build_face_cascade(flows, [f.execute(subs0) for f in flows], pipe, keep_fraction=0.7, stages=AWKWARD[sequence])
    assert len({id(s.flow) for s in stages if s.flow is not None}) >= 3
    dc = DeviceCascade(stages, (16, 16), K, pipe)
    got = dc.detect(torch.from_numpy(frame).cuda(), smallest_face=0.3)

    def extract(coords, dang):
        return pt.extract(frame, coords, (16, 16), dtype=np.uint8, delta_angs=dang) if len(coords) else np.zeros((0, 256), np.uint8)

    def execute(k, subs):
        return stages[k].flow.execute(subs)

    checked = []

    def regress(k, sl):
        clf = stages[k].classifier
        x = np.ascontiguousarray(sl[:, :clf.input_dim])
        dev_reg = clf.regression(x)
        want = ref_c.gauss_regression(x.astype(np.float64), clf.means, clf.inv_covs, clf._sqrt_def_covs, clf.p, clf.avg_labels, want_std=False)
        assert np.allclose(dev_reg, want, rtol=1e-9, atol=1e-9 * max(np.abs(want).max(), 1e-300)), (k, stages[k].name)
        checked.append((k, len(x)))
        return dev_reg              # the loop is fed the device value so that coordinates can be compared bit for bit
    ref = CR.run_cascade([s.name for s in stages], [s.flow is not None for s in stages], boxes, level, pipe, extract, execute, regress)
    assert [k for k, _ in checked] == list(range(len(stages))) and all(n > 0 for _, n in checked)     # every stage, on real rows
    # survivor counts are read back (a poll of a pinned word) after every Disc stage and at the end (-1 elsewhere: the count stays
    # on the device)
    known = [i for i, c in enumerate(got["counts"]) if c >= 0 and stages[i].type == "Disc"]
    assert len(known) == sum(s.type == "Disc" for s in stages) and known[0] == 0
    assert [got["counts"][i] for i in known] == [ref["counts"][i] for i in known], (got["counts"], ref["counts"])
    assert got["rows_executed"] >= ref["rows_executed"]        # launches between two Disc stages are sized by the last count read
    assert 0 < got["counts"][-1] < len(boxes) and got["counts"][0] < len(boxes)
    assert np.array_equal(got["orig_index"], ref["orig_index"])
    assert np.array_equal(got["coords"], ref["coords"]) and np.array_equal(got["angles"], ref["angles"])
    assert np.allclose(got["confidence"], ref["confidence"], rtol=1e-12, atol=1e-12)
    if any(s.type == "PAng" for s in stages[:-1]):
        assert np.abs(got["angles"]).max() > 0       # rotated extraction really took part
    # stage groups against one launch pair per stage (HIGSFA_CASCADE_NO_GROUPS, read per frame), and the device-computed grid
    # against windows handed over by the host: the same survivors with the same bits, the same counts wherever both report one
    import os
    os.environ["HIGSFA_CASCADE_NO_GROUPS"] = "1"
    try:
        single = dc.detect(torch.from_numpy(frame).cuda(), smallest_face=0.3, windows=(boxes, level))
    finally:
        del os.environ["HIGSFA_CASCADE_NO_GROUPS"]
    for key in ("coords", "angles", "orig_index", "confidence"):
        assert np.array_equal(single[key], got[key]), key
    assert all(a == b for a, b in zip(single["counts"], got["counts"]) if a >= 0 and b >= 0) and single["rows_executed"] == got["rows_executed"]
    assert sum(c >= 0 for c in got["counts"]) >= sum(c >= 0 for c in single["counts"])      # a group reports every stage's count
    assert [c for c in got["counts"] if c >= 0] == [ref["counts"][i] for i, c in enumerate(got["counts"]) if c >= 0]
    # several frames' worth of windows in ONE batch (the glue kernel was one workgroup): the frame's windows repeated
    # until there are more than 20 000 — first stages on the multi-workgroup glue (mark + scatter), later ones on the single
    # workgroup — must give the single batch's survivors once per copy, in order, bit for bit
    copies = -(-20500 // len(boxes))
    big = dc.detect(torch.from_numpy(frame).cuda(), smallest_face=0.3, windows=(np.tile(boxes, (copies, 1)), np.tile(level, (copies, 1))))
    assert big["n_windows"] == copies * len(boxes) > 20000
    n1 = len(got["orig_index"])
    assert len(big["orig_index"]) == copies * n1
    assert np.array_equal(big["orig_index"], (np.arange(copies)[:, None] * len(boxes) + got["orig_index"][None, :]).reshape(-1))
    assert np.array_equal(big["coords"], np.tile(got["coords"], (copies, 1))) and np.array_equal(big["angles"], np.tile(got["angles"], copies))
    assert np.array_equal(big["confidence"], np.tile(got["confidence"], copies))
    assert [c for c in big["counts"] if c >= 0] == [copies * c for c in got["counts"] if c >= 0]
    # the pieces the loop was fed: windows vs PIL's rule, features vs the oracle
    assert np.array_equal(subs0[::7], pil_restate.extract_subimages_rotate(frame, boxes[::7], np.zeros(len(boxes[::7])), (16, 16)))
    r = mdp_restate.execute_flow(nodes, subs0[::5])
    assert np.abs(feats[::5] - r).max() <= 1e-4 * np.abs(r).max()
    dc.close()
    for f in flows:
        f.close()


@pytest.mark.gpu
def test_device_grid_equals_host_grid(native_lib):
    """hg_cascade_grid_device (the first-stage windows from the grid's closed form, face_analysis.py:630-669) against
    grid.frame_boxes / cascade.frame_windows — numpy.linspace and the reference's box formula — bit for bit, on the configs[2]
    frame, on the shipped pipeline's 64x64 windows, on a frame barely larger than one window and on a single-column grid."""
    import ctypes as C
    import torch
    from pyfaceanalysis_amd import _capi
    from pyfaceanalysis_amd.cascade import frame_levels, frame_windows
    L = _capi.lib()
    cases = [(1000, 562, 0.1, (128, 128)), (1000, 562, 0.2, (64, 64)), (1920, 1080, 0.05, (64, 64)), (160, 90, 0.3, (16, 16)),
             (133, 131, 0.9, (128, 128)), (40, 400, 0.5, (32, 32)), (3648, 2736, 0.02, (64, 64))]
    for fw, fh, sf, sub in cases:
        boxes, level = frame_windows(fw, fh, sf, grid.FACE_PIPELINE, sub)
        levels, n_levels, n0 = frame_levels(fw, fh, sf, grid.FACE_PIPELINE, sub)
        assert n0 == len(boxes) > 0, (fw, fh, sf, sub)
        n = C.c_int64()
        _capi.check(L.hg_cascade_grid_device(0, levels, n_levels, None, None, 0, C.byref(n), None))
        assert n.value == n0
        b_dev = torch.empty((n0, 4), dtype=torch.float64, device="cuda")
        l_dev = torch.empty((n0, 3), dtype=torch.float64, device="cuda")
        _capi.check(L.hg_cascade_grid_device(0, levels, n_levels, C.c_void_p(b_dev.data_ptr()), C.c_void_p(l_dev.data_ptr()), n0, C.byref(n),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert np.array_equal(b_dev.cpu().numpy(), boxes), (fw, fh, sf, sub)
        assert np.array_equal(l_dev.cpu().numpy(), level), (fw, fh, sf, sub)
    with pytest.raises(ValueError):
        _capi.check(L.hg_cascade_grid_device(0, levels, 0, None, None, 0, C.byref(n), None))


@pytest.mark.gpu
def test_multi_classifier_regression_equals_single_launches(native_lib):
    """hg_gauss_regression_multi_device (the regressions of a stage group in one launch) against one
    hg_gauss_regression_device call per classifier: the same bits, for the class / width mix of the pipeline's groups
    (10 x 9 Disc, 50 x 10 / 50 x 20 pose regressors), below and above the four-rows-per-workgroup threshold, F32 and F64 rows."""
    import ctypes as C
    import torch
    from pyfaceanalysis_amd import _capi, synth_cascade
    L = _capi.lib()
    rng = np.random.default_rng(5)
    feats = rng.normal(size=(600, 20))
    clfs = [synth_cascade.quantile_classifier(feats, 10, np.linspace(-1, 1, 50)), synth_cascade.quantile_classifier(feats, 20, np.linspace(-3, 2, 50)),
            synth_cascade.quantile_classifier(feats, 9, (np.arange(10) + 0.5) / 10), synth_cascade.quantile_classifier(feats, 20, np.linspace(0.8, 0.85, 50))]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (1, 7, 255, 256, 1031):
        for dt, code in ((torch.float32, _capi.HG_F32), (torch.float64, _capi.HG_F64)):
            x = torch.from_numpy(rng.normal(size=(n, 24))).to(dt).cuda()
            for m in (1, 2, 4):
                hs = (C.c_void_p * m)(*[c._handle(c.avg_labels) for c in clfs[:m]])
                multi = torch.full((m, n + 5), np.nan, dtype=torch.float64, device="cuda")
                _capi.check(L.hg_gauss_regression_multi_device(hs, m, C.c_void_p(x.data_ptr()), code, n, 24, C.c_void_p(multi.data_ptr()), n + 5, st))
                for s in range(m):
                    one = torch.empty(n, dtype=torch.float64, device="cuda")
                    _capi.check(L.hg_gauss_regression_device(hs[s], C.c_void_p(x.data_ptr()), code, n, 24, C.c_void_p(one.data_ptr()), None, st))
                    torch.cuda.synchronize()
                    assert np.array_equal(multi[s, :n].cpu().numpy(), one.cpu().numpy()), (n, dt, m, s)
                    assert bool(torch.isnan(multi[s, n:]).all())
    hs = (C.c_void_p * 5)(*[clfs[0]._handle(clfs[0].avg_labels)] * 5)
    with pytest.raises(ValueError):
        _capi.check(L.hg_gauss_regression_multi_device(hs, 5, C.c_void_p(x.data_ptr()), code, 1, 24, C.c_void_p(multi.data_ptr()), 8, st))
    for c in clfs:
        c.close()


@pytest.mark.gpu
def test_config3_full_pyramid_1080p(native_lib, nets):
    """BASELINE.json configs[2]: one synthetic 1920x1080 frame, smallest_face = 0.1, prescaled to 1000x562 like the
    reference (FaceDetectUpdated.py:551-556) -> 10 pyramid levels, 1738 first-stage windows, cut at 128x128
    on the device and pushed through the 11-layer net in ONE execute; then the whole synthetic 17-stage cascade on the
    device.  Checked: prescale and EVERY window bit-exact against PIL; features against the oracle on a slice; the
    chained cascade against the restated stage loop (same survivors, coordinates, angles)."""
    import torch
    from PIL import Image
    from oracle import mdp_restate
    from pyfaceanalysis_amd import synth, synth_cascade
    from pyfaceanalysis_amd.cascade import DeviceCascade, frame_windows
    from pyfaceanalysis_amd.flow import Flow
    from pyfaceanalysis_amd.patches import Patcher
    nodes = nets("U11L-128")
    flow = Flow(nodes, output_dtype=np.float32)
    rng = np.random.default_rng(synth.INPUT_SEED)
    frame = np.rint(synth._box3(rng.integers(0, 256, (1080, 1920), dtype=np.uint8))).astype(np.uint8)
    pipe = dict(grid.FACE_PIPELINE)
    stages0 = [synth_cascade.Stage("Disc1", flow, synth_cascade.quantile_classifier(rng.normal(size=(50, 20)), 9, [0.0, 1.0]))]
    dc0 = DeviceCascade(stages0, (128, 128), 20, pipe)
    fdev = torch.from_numpy(frame).cuda()
    small_dev = dc0.prescale(fdev)
    small = small_dev.cpu().numpy()
    assert small.shape == (562, 1000)
    assert np.array_equal(small, np.asarray(Image.fromarray(frame, "L").resize((1000, 562), Image.NEAREST)))
    boxes, level = frame_windows(1000, 562, 0.1, pipe, (128, 128))
    assert len(boxes) == 1738 and len(np.unique(level[:, 2])) == 10
    pt = Patcher()
    subs = pt.extract(small, boxes, (128, 128), dtype=np.uint8)
    im = Image.fromarray(small, "L")
    for i, b in enumerate(boxes):                 # every window against PIL
        assert np.array_equal(subs[i], np.asarray(im.transform((128, 128), Image.EXTENT, tuple(b), Image.NEAREST)).reshape(-1)), i
    feats = flow.execute(subs, n_cols=20)         # all ten levels in one call
    idx = np.arange(0, 1738, 41)
    ref = mdp_restate.execute_flow(nodes, subs[idx])[:, :20]
    assert np.abs(feats[idx] - ref).max() <= 1e-4 * np.abs(ref).max()
    # the cascade, device resident, against the restated loop fed with the same device features
    stages = synth_cascade.build_face_cascade(flow, feats, pipe, keep_fraction=0.1)
    dc = DeviceCascade(stages, (128, 128), 20, pipe)
    got = dc.detect(small_dev, smallest_face=0.1)
    assert got["n_windows"] == 1738 and got["counts"][0] < 600 and got["rows_executed"] >= 1738
    # prescale + grid + stage loop as one host call (what bench.py times): the same answer from the full-size frame
    whole = dc.detect_frame(fdev, smallest_face=0.1)
    assert whole["n_windows"] == 1738 and list(whole["counts"]) == list(got["counts"]) and whole["rows_executed"] == got["rows_executed"]
    for key in ("coords", "angles", "orig_index", "confidence"):
        assert np.array_equal(whole[key], got[key]), key
    # another frame of the same size: the index tables of the prescale and of the first-stage grid are reused (they depend on sizes
    # only, hg_patcher_extract_keyed_device) — same answer as the path that rebuilds them from explicit windows
    frame2 = np.ascontiguousarray(frame[::-1, ::-1])
    f2dev = torch.from_numpy(frame2).cuda()
    again = dc.detect_frame(f2dev, smallest_face=0.1)
    small2 = dc.prescale(f2dev)
    assert np.array_equal(small2.cpu().numpy(), np.asarray(Image.fromarray(frame2, "L").resize((1000, 562), Image.NEAREST)))
    explicit = dc.detect(small2, smallest_face=0.1, windows=(boxes, level))
    assert list(again["counts"]) == list(explicit["counts"]) and again["counts"] != got["counts"]
    for key in ("coords", "angles", "orig_index", "confidence"):
        assert np.array_equal(again[key], explicit[key]), key
    assert np.array_equal(dc.detect_frame(fdev, smallest_face=0.1)["coords"], got["coords"])

    def extract(coords, dang):
        return pt.extract(small, coords, (128, 128), dtype=np.uint8, delta_angs=dang) if len(coords) else np.zeros((0, 16384), np.uint8)

    def execute(k, s):
        return flow.execute(s, n_cols=20)

    def regress(k, sl):
        return stages[k].classifier.regression(np.ascontiguousarray(sl[:, :stages[k].classifier.input_dim]))
    want = CR.run_cascade([s.name for s in stages], [s.flow is not None for s in stages], boxes, level, pipe, extract, execute, regress)
    # (the host loop and the device loop compute in different precisions between the network calls: decisions agree, numbers to 1e-3)
    known = [i for i, c in enumerate(got["counts"]) if c >= 0]
    assert [got["counts"][i] for i in known] == [want["counts"][i] for i in known], (got["counts"], want["counts"])
    assert np.array_equal(got["orig_index"], want["orig_index"])
    assert np.allclose(got["coords"], want["coords"], rtol=0, atol=1e-3) and np.allclose(got["angles"], want["angles"], rtol=0, atol=1e-3)
    # several frames in flight (bench.py `frames_in_flight`): cascades with their own flow handles, one stream and one host
    # thread each, at the same time — every one of them must return exactly what the cascade returns on its own
    import threading
    others = []
    for _ in range(2):
        f2 = Flow(nodes, output_dtype=np.float32)
        others.append((f2, DeviceCascade(synth_cascade.build_face_cascade(f2, feats, pipe, keep_fraction=0.1), (128, 128), 20, pipe)))
    cascades = [dc] + [c for _, c in others]
    streams = [torch.cuda.Stream() for _ in cascades]
    results = [None] * len(cascades)

    def worker(i):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[i]):
            for _ in range(4):
                results[i] = cascades[i].detect(small_dev, smallest_face=0.1)
        streams[i].synchronize()
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(cascades))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for r in results:
        assert list(r["counts"]) == list(got["counts"]) and np.array_equal(r["orig_index"], got["orig_index"])
        assert np.array_equal(r["coords"], got["coords"]) and np.array_equal(r["angles"], got["angles"])
    for f2, c2 in others:
        c2.close()
        f2.close()
    dc.close()
    dc0.close()
    flow.close()
