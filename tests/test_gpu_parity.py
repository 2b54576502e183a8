"""GPU parity: HIP path (through the C ABI) vs the float64 oracle on identical batches.
Tolerance (BASELINE.json north_star): max|y - ref| / max|ref| <= 1e-4 in fp32."""
import glob
import os

import numpy as np
import pytest

from oracle import mdp_restate as oracle
from oracle import ref_c
from pyfaceanalysis_amd import _capi, synth
from pyfaceanalysis_amd.blob import blob_to_flow
from pyfaceanalysis_amd.classifier import GaussianClassifier
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-4            # relative to max|ref| (north_star)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_err(y, ref):
    return float(np.abs(np.asarray(y, dtype=np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("preset,kw", [
    ("T3L-8", {}), ("T5L-16", {}), ("T5L-16", {"layout": "separate"}), ("T5L-16", {"node_kind": "igsfa"}),
])
def test_small_nets_match_oracle(native_lib, nets, preset, kw, force_generic):
    nodes = nets(preset, **kw)
    side = synth.preset_input_side(preset)
    x = synth.make_subimages(67, side, dtype=np.float64)       # ragged: not a multiple of 16
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes, force_generic=force_generic)
    y = flow.execute(x)
    assert y.dtype == np.float64 and y.shape == ref.shape
    assert rel_err(y, ref) <= TOL
    flow.close()


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "flow_*.npz"))))
def test_golden_fixtures(native_lib, path, force_generic):
    """Committed fixtures (uint8 sub-images in, float64 features out)."""
    g = np.load(path)
    flow = Flow.from_blob(g["blob"].tobytes(), force_generic=force_generic)
    assert rel_err(flow.execute(g["x"]), g["y"]) <= TOL
    flow.close()


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("maker", [helpers.overlapping_net, helpers.linear_net, helpers.product_net, helpers.wide_merge_net])
def test_awkward_structures(native_lib, maker, force_generic):
    """Overlapping fields, irregular switchboards, uneven node widths, sel_exp, clone layers, folded
    affines, product expansions / Head / Cutoff (generic plan)."""
    nodes = maker(5)
    x = np.random.default_rng(2).normal(size=(53, nodes[0].input_dim)) * 1.5
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes, force_generic=force_generic)
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzzed_hierarchies(native_lib, seed):
    """Random hierarchies (helpers.fuzz_net): random grids, overlaps, merge directions, uneven widths, clone and
    linear layers, odd exponents, sel_exp — fused plan and generic plan against the oracle, ragged batch sizes."""
    nodes = helpers.fuzz_net(seed)
    n = [1, 15, 16, 17, 33, 100][seed % 6]
    x = np.random.default_rng(seed).normal(size=(n, nodes[0].input_dim)) * 1.5
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    assert flow.info().plan_kind == _capi.HG_PLAN_FUSED
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()
    if seed % 4 == 0:
        flow = Flow(nodes, force_generic=True)
        assert rel_err(flow.execute(x), ref) <= TOL
        flow.close()


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_product_hierarchies(native_lib, seed):
    """Hierarchies with cross-column product expansions (QT, pair_prodsadj*, sel_exp of them — the vocabulary behind the
    reference's "Non-Linear" networks, FaceDetectUpdated.py:57,62), CutoffNode after the expansion, HeadNode after the
    node: fused plan (table-driven stage kernel) against the oracle; every fourth seed also the generic plan."""
    nodes = helpers.fuzz_product_net(seed)
    n = [1, 16, 17, 33, 100, 300][seed % 6]
    x = np.random.default_rng(seed).normal(size=(n, nodes[0].input_dim)) * 1.2
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    assert flow.info().plan_kind == _capi.HG_PLAN_FUSED and "table-driven expansion" in flow.describe()
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()
    if seed % 4 == 0:
        flow = Flow(nodes, force_generic=True)
        assert rel_err(flow.execute(x), ref) <= TOL
        flow.close()


def test_product_hierarchy_both_plans(native_lib):
    """A hierarchy of realistic size with product expansions in every layer (64x64 input, 256 first-layer nodes) at
    N = 4096: the fused plan (k_stage_prod) and the generic plan of the same flow agree to fp32 rounding, and both match the
    oracle.  (Their speed ratio is a measurement, not a parity property: tools/prod_stage_times.py, profiles/r02_product_plan.txt.)"""
    nodes = helpers.product_hier_net(1)
    n = 4096
    x = np.random.default_rng(0).normal(size=(n, nodes[0].input_dim)).astype(np.float32)
    res = {}
    for name, force in (("fused", False), ("generic", True)):
        flow = Flow(nodes, output_dtype=np.float32, force_generic=force)
        assert flow.info().plan_kind == (_capi.HG_PLAN_GENERIC if force else _capi.HG_PLAN_FUSED)
        res[name] = flow.execute(x)
        flow.close()
    assert rel_err(res["fused"], res["generic"].astype(np.float64)) <= TOL
    assert rel_err(res["fused"][:64], oracle.execute_flow(nodes, x[:64])) <= TOL


@pytest.mark.parametrize("kw", [{}, {"node_kind": "igsfa"}])
def test_non_finite_rows_stay_in_their_row(native_lib, nets, kw):
    """A NaN / Inf pixel poisons only its own sub-image (rows are independent): the other rows are
    bit-identical to a clean run, the poisoned rows come back non-finite, nothing crashes."""
    nodes = nets("T5L-16", **kw)
    x = synth.make_subimages(40, 16, dtype=np.float64)
    flow = Flow(nodes)
    clean = flow.execute(x)
    bad = x.copy()
    bad[3, 17] = np.nan
    bad[21, 200] = np.inf
    y = flow.execute(bad)
    keep = np.ones(40, dtype=bool)
    keep[[3, 21]] = False
    assert np.array_equal(y[keep], clean[keep])
    assert not np.isfinite(y[3]).all() and not np.isfinite(y[21]).all()
    flow.close()


@pytest.mark.parametrize("n", [5, 300])
def test_remainder_tiles(native_lib, n, monkeypatch):
    """Layers whose affines end in a tile of <= 4 real rows (helpers.remainder_net): the 4x4-MFMA form of k_stage
    (one / two batch tiles per wave) against the oracle, and against the ordinary form (HIGSFA_NO_REM4)."""
    nodes = helpers.remainder_net(3)
    x = np.random.default_rng(n).normal(size=(n, nodes[0].input_dim)) * 1.5
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    desc = flow.describe()
    assert desc.count("(4x4 remainder tiles)") == 2
    y = flow.execute(x)
    assert rel_err(y, ref) <= TOL
    flow.close()
    monkeypatch.setenv("HIGSFA_NO_REM4", "1")
    flow = Flow(nodes)
    assert "(4x4 remainder tiles)" not in flow.describe()
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()


@pytest.mark.parametrize("seed", range(12))
def test_fuzzed_igsfa_hierarchies(native_lib, seed, monkeypatch):
    """Random iGSFA hierarchies (helpers.fuzz_igsfa_net) against the oracle.  Nodes of up to 64 inputs are folded
    on the host into ordinary nodes; odd seeds switch the folding off (HIGSFA_IG_NOFOLD, read at plan time) so
    that the three-GEMM node kernel is exercised at every width; every fourth seed also runs the generic plan."""
    if seed % 2:
        monkeypatch.setenv("HIGSFA_IG_NOFOLD", "1")
    nodes = helpers.fuzz_igsfa_net(seed)
    n = [1, 16, 17, 50][seed % 4]
    x = np.random.default_rng(seed).normal(size=(n, nodes[0].input_dim)) * 1.5
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    assert flow.info().plan_kind == _capi.HG_PLAN_FUSED
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()
    if seed % 4 == 0:
        flow = Flow(nodes, force_generic=True)
        assert rel_err(flow.execute(x), ref) <= TOL
        flow.close()


@pytest.mark.parametrize("force_generic", [True, False])
def test_u11l_128_matches_oracle(native_lib, nets, force_generic):
    """BASELINE.json configs[0]: the 11-layer net on 256 sub-images of 128x128."""
    nodes = nets("U11L-128")
    x = synth.make_subimages(256, 128, dtype=np.float32)
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes, force_generic=force_generic)
    y = flow.execute(x)
    err = rel_err(y, ref)
    percol = np.abs(y - ref).max(axis=0)[:20] / np.abs(ref).max()
    print("U11L-128 generic=%s max|d|/max|ref| = %.3e; worst of first 20 cols %.3e" % (force_generic, err, percol.max()))
    assert err <= TOL
    flow.close()


def test_u11l_128_igsfa_matches_oracle(native_lib, nets):
    """The 11-layer hierarchy with iGSFA nodes (HiGSFA proper) on the fused plan:
    gather + 11 iGSFA stages, three chained GEMMs per node in registers."""
    nodes = nets("U11L-128", node_kind="igsfa")
    x = synth.make_subimages(200, 128, dtype=np.uint8)
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    assert flow.info().plan_kind == _capi.HG_PLAN_FUSED
    y = flow.execute(x)
    err = rel_err(y, ref)
    print("U11L-128 iGSFA max|d|/max|ref| = %.3e" % err)
    assert err <= TOL
    gen = Flow(nodes, force_generic=True)
    assert rel_err(gen.execute(x[:48]), ref[:48]) <= TOL
    flow.close()
    gen.close()


def test_u11l_64_matches_oracle(native_lib, nets):
    """The shipped pipelines feed 64x64 sub-images (Pipelines/Pipeline_experimental.txt:2)."""
    nodes = nets("U11L-64")
    x = synth.make_subimages(100, 64, dtype=np.float64)
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    assert flow.info().plan_kind == _capi.HG_PLAN_FUSED
    assert rel_err(flow.execute(x), ref) <= TOL
    flow.close()


def test_linear_11_layer_net_on_96x96(native_lib):
    """The age pipeline's shape (Pipelines/Pipeline_experimental.txt:4,64; face_analysis.py:1257): a linear 11-layer network
    on 96x96 sub-images — a side that is not 4 * 2^k, 3x3 receptive fields, affine-only nodes — through the fused plan; the
    caller reads 4-5 features of it."""
    nodes = helpers.linear_u11l_96(2)
    flow = Flow(nodes)
    inf = flow.info()
    assert inf.plan_kind == _capi.HG_PLAN_FUSED and inf.input_dim == 96 * 96 and inf.n_top_nodes == 22
    x = synth.make_subimages(40, 96, dtype=np.float64)
    ref = oracle.execute_flow(nodes, x)
    assert rel_err(flow.execute(x), ref) <= TOL
    assert np.array_equal(flow.execute(x.astype(np.uint8), n_cols=5), flow.execute(x, n_cols=5))
    assert rel_err(flow.execute(x[:1]), ref[:1]) <= TOL          # one face at a time is how the reference calls it (:1257)
    gen = Flow(nodes, force_generic=True)
    assert rel_err(gen.execute(x), ref) <= TOL
    flow.close()
    gen.close()


def test_input_dtypes_layouts_and_edges(native_lib, nets):
    nodes = nets("T5L-16")
    flow = Flow(nodes)
    xi = synth.make_subimages(50, 16, dtype=np.uint8)
    ref = oracle.execute_flow(nodes, xi)
    y8 = flow.execute(xi)
    assert rel_err(y8, ref) <= TOL
    # integer pixels are exact in every input type -> identical device arithmetic
    assert np.array_equal(y8, flow.execute(xi.astype(np.float32)))
    assert np.array_equal(y8, flow.execute(xi.astype(np.float64)))
    assert np.array_equal(y8, flow.execute(np.asfortranarray(xi.astype(np.float64))))      # F order
    wide = np.zeros((50, 300), dtype=np.float32)
    wide[:, 7:263] = xi
    assert np.array_equal(y8, flow.execute(wide[:, 7:263]))                                # strided rows, unaligned
    assert np.array_equal(y8, flow.execute(xi.astype(np.int32)))                           # other dtypes are cast
    # batch-size edges: the reference calls with N = 1..728, guards N == 0 itself
    for n in (0, 1, 15, 16, 17, 33):
        yn = flow.execute(xi[:n])
        assert yn.shape == (n, 10)
        assert np.array_equal(yn, y8[:n])                   # a row's result never depends on its neighbours
    # first-k columns (the caller consumes sl[:, 0:reg_num_signals], FaceDetectUpdated.py:719)
    assert np.array_equal(flow.execute(xi, n_cols=4), y8[:, :4])
    f32 = Flow(nodes, output_dtype=np.float32)
    y32 = f32.execute(xi)
    assert y32.dtype == np.float32 and np.array_equal(y32.astype(np.float64), y8)
    with pytest.raises(_capi.NodeException):
        flow.execute(np.zeros((4, 255)))
    flow.close()
    f32.close()


def test_nodenr_and_benchmark_kwarg(native_lib, nets):
    nodes = nets("T5L-16")
    flow = Flow(nodes)
    x = synth.make_subimages(20, 16, dtype=np.float64)
    for nodenr in (0, 1, 3, 9):
        ref = oracle.execute_flow(nodes, x, nodenr=nodenr)
        assert rel_err(flow.execute(x, nodenr=nodenr), ref) <= TOL

    class Bench(object):                      # interface of benchmarking.Benchmark (benchmarking.py:39-58)
        enabled = True
        default_reference = "networks"

        def __init__(self):
            self.tasks = []

        def add_task_ellapsed(self, label, secs, reference=None):
            self.tasks.append((reference, label, secs))

    b = Bench()
    flow.execute(x, benchmark=b)
    assert len(b.tasks) == flow.info().n_stages and all(t[0] == "networks" and t[2] >= 0 for t in b.tasks)
    flow.close()


def test_full_size_properties(native_lib, nets):
    """BASELINE.json configs[1] size (4096 x 128x128), where the oracle is too slow to be the only check:
    (1) fused and generic plans (independent HIP implementations) agree; (2) determinism: two runs are
    bit-identical; (3) row-permutation equivariance, bit-exact: a sub-image's features do not depend on
    where in the batch it sits; (4) a slice is checked against the oracle."""
    nodes = nets("U11L-128")
    n = 4096
    x = synth.make_subimages(n, 128, dtype=np.uint8)
    fused, generic = Flow(nodes, output_dtype=np.float32), Flow(nodes, output_dtype=np.float32, force_generic=True)
    y = fused.execute(x, n_cols=20)
    assert np.array_equal(y, fused.execute(x, n_cols=20))
    yg = generic.execute(x, n_cols=20)
    assert rel_err(y, yg.astype(np.float64)) <= TOL
    perm = np.random.default_rng(0).permutation(n)
    assert np.array_equal(fused.execute(x[perm], n_cols=20), y[perm])
    idx = np.arange(0, n, 64)
    ref = oracle.execute_flow(nodes, x[idx])[:, :20]
    assert rel_err(y[idx], ref) <= TOL
    fused.close()
    generic.close()


def test_gaussian_regression(native_lib):
    """hg_gauss_regression on the parameters of seven of the reference's own classifier files — one per (classes, features)
    shape of SavedClassifiers/*.pckl, the K = 50, d = 20 pose regressors (sqrt-determinants ~1e42) included; rows far from
    every class and rows exactly on a class mean are part of the fixture (FaceDetectUpdated.py:709-719)."""
    g = np.load(os.path.join(GOLD, "classifiers.npz"))
    assert int(g["n_classifiers"]) == 7
    for i in range(7):
        clf = GaussianClassifier(g["c%d_means" % i], g["c%d_inv_covs" % i], g["c%d_sqrt_def_covs" % i], g["c%d_p" % i],
                                 avg_labels=g["c%d_avg_labels" % i])
        x = g["c%d_x" % i]
        reg, std = clf.regression(x, estimate_std=True)
        assert np.allclose(reg, g["c%d_reg" % i], rtol=1e-9, atol=1e-9 * np.abs(g["c%d_reg" % i]).max())
        assert np.allclose(std, g["c%d_std" % i], rtol=1e-6, atol=1e-7 * (1 + np.abs(g["c%d_std" % i]).max()))
        reg32 = clf.regression(x.astype(np.float32))
        ref32 = ref_c.gauss_regression(x.astype(np.float32).astype(np.float64), clf.means, clf.inv_covs, clf._sqrt_def_covs,
                                       clf.p, clf.avg_labels, want_std=False)
        assert np.allclose(reg32, ref32, rtol=1e-9, atol=1e-9 * np.abs(ref32).max())
        assert clf.regression(x[:0]).shape == (0,)
        # the one-wave-per-row kernel (classifiers too large for the workgroup form's LDS) gives the same bits
        os.environ["HIGSFA_GAUSS_WAVE"] = "1"
        try:
            reg_w, std_w = clf.regression(x, estimate_std=True)
        finally:
            del os.environ["HIGSFA_GAUSS_WAVE"]
        assert np.array_equal(reg_w, reg) and np.array_equal(std_w, std)
        # from 256 rows on a workgroup takes four rows: the same bits row by row, ragged last workgroup included
        reps = -(-301 // len(x))
        xx = np.tile(x, (reps, 1))[:301]
        reg4, std4 = clf.regression(xx, estimate_std=True)
        assert np.array_equal(reg4, np.tile(reg, reps)[:301]) and np.array_equal(std4, np.tile(std, reps)[:301])
        clf.close()
    # a classifier beyond the workgroup form (K d > 4096): 300 classes x 20 features, against the C restatement
    rng = np.random.default_rng(5)
    K, d = 300, 20
    means = rng.normal(size=(K, d)) * 3
    A = rng.normal(size=(K, d, d))
    inv_covs = A @ A.transpose(0, 2, 1) / d + np.eye(d) * 0.5
    sqrt_def = np.array([np.sqrt(np.linalg.det(np.linalg.inv(ic))) for ic in inv_covs])
    clf = GaussianClassifier(means, inv_covs, sqrt_def, np.full(K, 1.0 / K), avg_labels=rng.normal(size=K))
    x = means[rng.integers(0, K, 40)] + rng.normal(size=(40, d)) * 0.3
    want = ref_c.gauss_regression(x, clf.means, clf.inv_covs, clf._sqrt_def_covs, clf.p, clf.avg_labels, want_std=False)
    assert np.allclose(clf.regression(x), want, rtol=1e-9, atol=1e-9 * np.abs(want).max())
    clf.close()
