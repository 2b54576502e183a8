"""CPU: the C-ABI library loads, exports every symbol include/higsfa.h declares, parses and plans
flows on the host, rejects malformed blobs, and fails loudly without a GPU (no compute calls)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from pyfaceanalysis_amd import _capi, blob, synth
from pyfaceanalysis_amd import nodes as N
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(lib, data, flags=0):
    h = C.c_void_p()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    rc = lib.hg_flow_load(buf, len(data), flags, C.byref(h))
    return rc, h


def test_header_symbols_exported(native_lib):
    text = open(os.path.join(ROOT, "include", "higsfa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(hg_[a-z_0-9]+)\s*\(", text)))
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(native_lib, name), "libhigsfa.so does not export %s" % name
    assert set(declared) == set(_capi.EXPORTED_SYMBOLS)
    assert native_lib.hg_version() == 100


def test_blob_roundtrip_is_stable():
    for maker in (helpers.overlapping_net, helpers.linear_net, helpers.product_net):
        nodes = maker(1)
        b1 = blob.flow_to_blob(nodes)
        b2 = blob.flow_to_blob(blob.blob_to_flow(b1))
        assert b1 == b2 and len(b1) % 8 == 0


def test_plans(native_lib, nets, monkeypatch):
    inf, desc = Flow(nets("T5L-16")).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and inf.input_dim == 256 and inf.output_dim == 10
    assert inf.flops_per_row == synth.flops_per_row(nets("T5L-16")) and inf.padded_flops_per_row >= inf.flops_per_row
    assert Flow(nets("T5L-16", layout="separate")).host_plan()[0].plan_kind == _capi.HG_PLAN_FUSED
    assert Flow(helpers.overlapping_net()).host_plan()[0].plan_kind == _capi.HG_PLAN_FUSED
    assert Flow(helpers.linear_net()).host_plan()[0].plan_kind == _capi.HG_PLAN_FUSED
    # iGSFA nodes of up to 64 inputs are folded into ordinary nodes on the host ...
    inf, desc = Flow(nets("T5L-16", node_kind="igsfa")).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and "iGSFA stage" not in desc and "fused gather" not in desc
    assert inf.padded_flops_per_row >= inf.flops_per_row > 0
    # ... unless told not to: gather pre-pass + three-GEMM node kernel
    monkeypatch.setenv("HIGSFA_IG_NOFOLD", "1")
    inf, desc = Flow(nets("T5L-16", node_kind="igsfa")).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and "fused gather" in desc and "fused iGSFA stage" in desc
    assert "folded" not in desc
    monkeypatch.delenv("HIGSFA_IG_NOFOLD")
    # product expansions (QT, pair products), HeadNode, CutoffNode after the expansion: table-driven fused stage
    inf, desc = Flow(helpers.product_net()).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and "table-driven expansion: products, clip" in desc and "fused gather" in desc
    inf, desc = Flow(helpers.fuzz_product_net(3)).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and "table-driven expansion" in desc
    assert Flow(nets("T5L-16"), force_generic=True).host_plan()[0].plan_kind == _capi.HG_PLAN_GENERIC
    inf, desc = Flow(helpers.wide_merge_net()).host_plan()      # too much weight per node for LDS -> generic, with a reason
    assert inf.plan_kind == _capi.HG_PLAN_GENERIC and "LDS" in desc


def test_u11l_128_plan(native_lib, nets):
    inf, desc = Flow(nets("U11L-128")).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and inf.n_stages == 12
    # issued work (round 5: <= 4-row tiles counted as the v_mfma_f32_4x4x1 they run on, 512 FLOP, not as 16 x 16 tiles: profiles/r05_issue_table.md)
    assert inf.flops_per_row == 11017088 and inf.padded_flops_per_row == 12158976
    assert "issued: 24576 x 16x16x4 + 0 x 4x4x1" in desc and "issued: 9216 x 16x16x4 + 9216 x 4x4x1" in desc
    assert inf.input_dim == 16384 and inf.output_dim == 60 and inf.n_top_nodes == 22
    # short batches: runs of layers that fall into independent sub-trees, one launch each (round 5; planned on the host: k_subtree) —
    # layers 6-8 under four roots, layers 3-5 under 32, and the alternative set one layer lower (5-7 under eight)
    assert "this and the next 2 layer(s) as 4 sub-trees in ONE launch" in desc and "this and the next 2 layer(s) as 32 sub-trees in ONE launch" in desc
    assert "3 layers from here as 8 sub-trees]" in desc and desc.count("[in the sub-tree launch for short batches]") == 4
    # iGSFA variant: three ordinary (folded) layers, then wide nodes on the node kernel in its folded form
    inf, desc = Flow(nets("U11L-128", node_kind="igsfa")).host_plan()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and inf.n_stages == 12
    assert desc.count("fused iGSFA stage (folded to one GEMM)") == 8 and "fused gather" not in desc


def test_subtree_runs_are_planned_where_layers_split(native_lib):
    """plan_subtree on the host: hierarchies without overlap get runs of sub-trees (4 ... 32 roots, merges of 2, 3, 4 children); a
    net whose receptive fields overlap gets none."""
    planned = 0
    for seed in range(16):
        _, desc = Flow(helpers.subtree_fuzz_net(seed)).host_plan()
        for ln in desc.splitlines():
            if "sub-trees in ONE launch" in ln:
                assert int(ln.split(" sub-trees in ONE launch")[0].split(" as ")[-1]) >= 4
        planned += "sub-trees in ONE launch" in desc
    assert planned >= 12
    _, desc = Flow(helpers.overlapping_net(5)).host_plan()
    assert "sub-trees in ONE launch" not in desc


def test_malformed_blobs_are_rejected(native_lib):
    good = blob.flow_to_blob(helpers.overlapping_net())
    rc, h = _load(native_lib, good)
    assert rc == 0
    native_lib.hg_flow_free(h)
    bad = [good[:10], good[:40], good[:len(good) // 2], good[:-8], b"XXXXXXXX" + good[8:],
           good[:8] + struct.pack("<I", 9) + good[12:], good + b"\0" * 8]
    # corrupt the first switchboard connection (header 24 + flow head 16 + node head 16 = offset 56)
    ba = bytearray(good)
    struct.pack_into("<i", ba, 56, 10 ** 6)
    bad.append(bytes(ba))
    ba = bytearray(good)
    struct.pack_into("<I", ba, 24 + 4, 77)      # flow input_dim no longer matches the first node
    bad.append(bytes(ba))
    rng = np.random.default_rng(0)
    for _ in range(40):                         # random single-word corruptions must never crash
        ba = bytearray(good)
        off = int(rng.integers(6, 200)) * 4
        struct.pack_into("<I", ba, off, int(rng.integers(0, 2 ** 32)))
        bad.append(bytes(ba))
    n_rejected = 0
    for b in bad:
        rc, h = _load(native_lib, b)
        if rc == 0:
            native_lib.hg_flow_free(h)          # a corrupted float payload is still a valid flow
        else:
            n_rejected += 1
            assert rc in (_capi.HG_ERR_FORMAT, _capi.HG_ERR_DIM, _capi.HG_ERR_ARG)
            assert len(native_lib.hg_last_error()) > 0
    assert n_rejected >= 9
    rc = native_lib.hg_flow_load(None, 0, 0, C.byref(C.c_void_p()))
    assert rc == _capi.HG_ERR_ARG


def test_malformed_expansion_records_are_rejected(native_lib):
    """Expansion selections / offsets the device would turn into out-of-range reads are refused at load time."""
    def net(func, p=8):
        rng = np.random.default_rng(0)
        ex = N.GeneralExpansionNode([N.identity, func], p)
        return [N.Layer([N.FlowNode([helpers.rand_pca(rng, 6, p), ex, helpers.rand_sfa(rng, ex.output_dim, 4)])])]

    def patched(func, **fields):
        good = blob.flow_to_blob(net(func))
        rec = struct.pack("<IIIId", blob._EXP_KIND[func.kind], func.sel, func.k, 0, func.expo)
        off = good.index(rec)
        kind, sel, k, expo = blob._EXP_KIND[func.kind], func.sel, func.k, func.expo
        kind, sel, k, expo = fields.get("kind", kind), fields.get("sel", sel), fields.get("k", k), fields.get("expo", expo)
        return good[:off] + struct.pack("<IIIId", kind, sel, k, 0, expo) + good[off + 24:]

    pp1, band2 = N.pair_prodsadj_ex(1, "offset"), N.pair_prodsadj_ex(2, "band")
    cases = [patched(pp1, k=0xFFFFFFFF),        # x_i * x_{i-1}: reads below the block
             patched(pp1, k=0), patched(pp1, k=8),
             patched(band2, k=0), patched(band2, k=9), patched(band2, k=0xFFFFFFFF), patched(band2, kind=6),
             patched(N.sel_exp(3, N.QT), sel=5),                 # selection changed, output width no longer matches
             patched(N.unsigned_08expo, expo=float("nan")), patched(N.unsigned_08expo, expo=-0.5),
             patched(N.signed_08expo, expo=float("inf"))]
    for b in cases:
        rc, h = _load(native_lib, b)
        assert rc in (_capi.HG_ERR_FORMAT, _capi.HG_ERR_DIM), native_lib.hg_last_error()
        assert b"expansion" in native_lib.hg_last_error()
    good = [blob.flow_to_blob(x) for x in (net(pp1), net(band2), net(N.pair_prodsadj_ex(8, "band")), net(N.QT), net(N.sel_exp(3, N.QT)), net(N.unsigned_expo(1.3)))]
    # a selection wider than the block reads all of it (numpy slicing clamps; cuicuilco's sel_exp): 9 of 8, 0xFFFFFFFE of 8
    good += [patched(N.QT, sel=9), patched(N.QT, sel=0xFFFFFFFE)]
    for b in good:
        rc, h = _load(native_lib, b)
        assert rc == 0, native_lib.hg_last_error()
        native_lib.hg_flow_free(h)


def test_igsfa_record_variants_load(native_lib):
    """The IGSFA record's flags (lr on scaled / unscaled features, per-column / matrix scaling) parse and plan on the
    host; unknown flag bits and a singular scaling with lr_input='unscaled' are refused."""
    for seed in range(6):
        nodes = helpers.fuzz_igsfa_net(seed)
        inf, _ = Flow(nodes).host_plan()
        assert inf.input_dim == nodes[0].input_dim and inf.output_dim == nodes[-1].output_dim
    nodes = helpers.fuzz_igsfa_net(1, lr_input="unscaled", scaling="per_column")
    for ig in nodes[1].nodes:
        ig.magn_n_sfa_x[0, 0] = 0.0
    if nodes[1].nodes[0].lr_node is not None:
        rc, h = _load(native_lib, blob.flow_to_blob(nodes))
        assert rc == _capi.HG_ERR_FORMAT and b"singular" in native_lib.hg_last_error()
    good = blob.flow_to_blob(helpers.fuzz_igsfa_net(0))
    off = good.index(struct.pack("<II", 1, 1)) if struct.pack("<II", 1, 1) in good else -1
    assert off > 0
    bad = good[:off] + struct.pack("<II", 1, 9) + good[off + 8:]
    rc, h = _load(native_lib, bad)
    assert rc == _capi.HG_ERR_FORMAT


def test_python_blob_validation():
    with pytest.raises(ValueError):
        blob.flow_to_blob([])
    with pytest.raises(ValueError):
        blob.flow_to_blob([N.IdentityNode(4), N.IdentityNode(5)])
    with pytest.raises(ValueError):
        N.Switchboard(4, [0, 4])
    with pytest.raises(ValueError):
        blob.blob_to_flow(b"nonsense" * 10)


def test_flow_container_protocol(nets):
    nodes = nets("T5L-16")
    f = Flow(nodes)
    assert len(f) == 10 and f[0] is nodes[0] and f.input_dim == 256 and f.output_dim == 10
    assert isinstance(f[:4], Flow) and len(f[:4]) == 4 and [n for n in f] == nodes
    with pytest.raises(_capi.NodeException):
        f.execute(np.zeros((3, 255)))            # dimension check happens before any device work
    with pytest.raises(_capi.NodeException):
        f.execute(np.zeros(256))
    with pytest.raises(ValueError):
        Flow([])


def test_sharded_entry_argument_checks(native_lib, nets):
    """hg_flow_execute_sharded validates like hg_flow_execute before touching any device, then needs one."""
    b = blob.flow_to_blob(nets("T3L-8"))
    rc, h = _load(native_lib, b)
    assert rc == 0
    x = np.zeros((4, 64))
    y = np.zeros((4, 6))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    call = lambda *a: native_lib.hg_flow_execute_sharded(h, *a)
    devs = (C.c_int * 2)(0, 0)
    assert call(vp(x), _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 6, 6, devs, 0) == _capi.HG_ERR_ARG          # no devices
    assert call(vp(x), _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 6, 6, devs, 65) == _capi.HG_ERR_ARG
    assert call(vp(x), _capi.HG_F64, 4, 63, vp(y), _capi.HG_F64, 6, 6, devs, 2) == _capi.HG_ERR_DIM           # ldx < input_dim
    assert call(vp(x), _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 7, 7, devs, 2) == _capi.HG_ERR_DIM           # y_cols > output_dim
    assert call(vp(x), 7, 4, 64, vp(y), _capi.HG_F64, 6, 6, devs, 2) == _capi.HG_ERR_ARG                      # dtype
    assert call(None, _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 6, 6, devs, 2) == _capi.HG_ERR_ARG            # null x
    assert call(vp(x), _capi.HG_F64, -1, 64, vp(y), _capi.HG_F64, 6, 6, devs, 2) == _capi.HG_ERR_ARG
    cnt = C.c_int(-1)
    native_lib.hg_device_count(C.byref(cnt))
    if cnt.value == 0:
        assert call(vp(x), _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 6, 6, devs, 2) == _capi.HG_ERR_DEVICE
        assert b"no HIP device" in native_lib.hg_last_error()
    else:
        bad = (C.c_int * 1)(cnt.value)
        assert call(vp(x), _capi.HG_F64, 4, 64, vp(y), _capi.HG_F64, 6, 6, bad, 1) == _capi.HG_ERR_DEVICE
    native_lib.hg_flow_free(h)


def test_no_gpu_means_loud_failure(native_lib, nets):
    cnt = C.c_int(-1)
    assert native_lib.hg_device_count(C.byref(cnt)) == 0
    if cnt.value > 0:
        pytest.skip("a GPU is visible; the no-device path cannot be exercised")
    f = Flow(nets("T3L-8"))
    with pytest.raises(RuntimeError, match="no HIP device"):
        f.execute(np.zeros((2, 64)))
    with pytest.raises(RuntimeError, match="no HIP device"):
        f.execute(np.zeros((2, 64)), devices=[0, 1])
    from pyfaceanalysis_amd.classifier import GaussianClassifier
    g = GaussianClassifier(np.zeros((2, 3)), np.stack([np.eye(3)] * 2), np.ones(2), np.ones(2) / 2, avg_labels=[0.0, 1.0])
    with pytest.raises(RuntimeError, match="no HIP device"):
        g.regression(np.zeros((2, 3)))


def test_ordering_event_entry_points_reject_null(native_lib):
    """hg_event_* (device-scope ordering events for pyfaceanalysis_amd/sharded.py): argument checks work without a GPU."""
    assert native_lib.hg_event_create(None) == _capi.HG_ERR_ARG
    assert native_lib.hg_event_record(None, None) == _capi.HG_ERR_ARG and b"null event" in native_lib.hg_last_error()
    assert native_lib.hg_stream_wait_event(None, None) == _capi.HG_ERR_ARG
    assert native_lib.hg_event_query(None) == _capi.HG_ERR_ARG
    native_lib.hg_event_destroy(None)          # a no-op
    import ctypes as C
    assert native_lib.hg_event_create_on(None, 0, 1) == _capi.HG_ERR_ARG
    h = C.c_void_p()
    rc = native_lib.hg_event_create_on(C.byref(h), 0, 0)
    if rc == _capi.HG_OK:                      # a box with a GPU
        native_lib.hg_event_destroy(h)
    else:
        assert rc == _capi.HG_ERR_DEVICE and b"no HIP device" in native_lib.hg_last_error()


def test_level_table_is_the_host_grid_and_round5_entries_check_their_arguments(native_lib):
    """Round 5, host side only (no GPU): the table of pyramid levels handed to hg_cascade_detect_levels_device describes exactly the
    windows grid.frame_boxes builds — numpy.linspace positions i * (stop / (n - 1)) with the end point exact, the reference's box
    formula (face_analysis.py:630-669) and the level constants, which is what k_cascade_init_grid evaluates on the device (bit for bit
    there: tests/test_cascade.py) — and hg_cascade_grid_device counts them without touching a device; the new entries reject bad
    arguments before any HIP call."""
    from pyfaceanalysis_amd import grid
    from pyfaceanalysis_amd.cascade import frame_levels, frame_windows
    L = _capi.lib()
    for fw, fh, sf, sub in ((1000, 562, 0.1, (128, 128)), (1000, 562, 0.2, (64, 64)), (160, 90, 0.3, (16, 16)), (133, 131, 0.9, (128, 128)), (40, 400, 0.5, (32, 32))):
        boxes, level = frame_windows(fw, fh, sf, grid.FACE_PIPELINE, sub)
        levels, n_levels, n0 = frame_levels(fw, fh, sf, grid.FACE_PIPELINE, sub)
        assert n0 == len(boxes)
        out_b, out_l = [], []
        for k in range(n_levels):
            v = levels[k]
            lin = lambda j, num, stop: 0.0 if (num <= 1 or j == 0) else (stop if j == num - 1 else j * (stop / (num - 1)))
            for iy in range(v.ny):
                for ix in range(v.nx):
                    x0, y0 = lin(ix, v.nx, v.x_stop), lin(iy, v.ny, v.y_stop)
                    out_b.append((x0, y0, x0 + v.patch_w - 1, y0 + v.patch_h - 1))
                    out_l.append((v.max_dx, v.max_dy, v.base_side))
        assert np.array_equal(np.array(out_b), boxes) and np.array_equal(np.array(out_l), level), (fw, fh, sf, sub)
        n = C.c_int64(-1)
        assert L.hg_cascade_grid_device(0, levels, n_levels, None, None, 0, C.byref(n), None) == _capi.HG_OK and n.value == n0
    assert L.hg_cascade_grid_device(0, levels, 0, None, None, 0, C.byref(n), None) == _capi.HG_ERR_ARG
    assert L.hg_cascade_grid_device(0, levels, 33, None, None, 0, C.byref(n), None) == _capi.HG_ERR_ARG
    levels[0].nx = 0
    assert L.hg_cascade_grid_device(0, levels, 1, None, None, 0, C.byref(n), None) == _capi.HG_ERR_ARG
    # the group regression: 1 .. 4 classifiers per launch, checked before anything else
    hs = (C.c_void_p * 5)()
    assert L.hg_gauss_regression_multi_device(hs, 5, None, _capi.HG_F32, 0, 20, None, 0, None) == _capi.HG_ERR_ARG
    assert L.hg_gauss_regression_multi_device(hs, 0, None, _capi.HG_F32, 0, 20, None, 0, None) == _capi.HG_ERR_ARG
    assert L.hg_gauss_regression_multi_device(hs, 2, None, _capi.HG_F32, 0, 20, None, 0, None) == _capi.HG_ERR_ARG      # null handles
    # the probes and the transport query
    t, d, tr = C.c_double(), C.c_int(), C.c_int()
    assert L.hg_host_pack_probe(None, _capi.HG_F64, 10, 16, 16, 1, C.byref(t)) == _capi.HG_ERR_ARG
    x = np.arange(64 * 48, dtype=np.float64).reshape(64, 48) % 256
    assert L.hg_host_pack_probe(x.ctypes.data_as(C.c_void_p), _capi.HG_F64, 64, 48, 48, 2, C.byref(t)) == _capi.HG_OK and 0 < t.value < 1.0
    assert L.hg_host_pack_probe(x.ctypes.data_as(C.c_void_p), _capi.HG_F64, 64, 40, 48, 2, C.byref(t)) == _capi.HG_ERR_ARG      # ldx < in_dim
    assert L.hg_host_store_probe(0, 1024, 1, C.byref(t), C.byref(d)) == _capi.HG_ERR_ARG
    assert L.hg_flow_host_transport(None, C.byref(tr)) == _capi.HG_ERR_ARG
