"""CPU: the oracle against golden fixtures, against the independent C restatement, and against
hand-computed known answers / properties of the node definitions."""
import glob
import os

import numpy as np
import pytest

from oracle import mdp_restate as oracle
from oracle import fast_cpu, ref_c
from pyfaceanalysis_amd import nodes as N
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.blob import blob_to_flow
from tests import helpers

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "flow_*.npz"))))
def test_golden_flows(path):
    g = np.load(path)
    nodes = blob_to_flow(g["blob"].tobytes())
    y = oracle.execute_flow(nodes, g["x"])
    assert np.abs(y - g["y"]).max() <= 1e-11 * np.abs(g["y"]).max()
    yc = ref_c.execute_flow(nodes, g["x"])
    assert np.abs(yc - g["y"]).max() <= 1e-11 * np.abs(g["y"]).max()


def test_fast_cpu_timing_leg_agrees():
    """oracle/fast_cpu (bench.py's "good CPU" timing point) computes the same flow: golden nets it covers
    to 1e-11, ragged row counts, 1 and 3 threads; node kinds it does not cover raise."""
    covered = 0
    for path in sorted(glob.glob(os.path.join(GOLD, "flow_*.npz"))):
        g = np.load(path)
        nodes = blob_to_flow(g["blob"].tobytes())
        try:
            plan = fast_cpu.Plan(nodes)
        except TypeError:
            continue
        covered += 1
        for threads in (1, 3):
            y = plan.run(g["x"], threads)
            assert np.abs(y - g["y"]).max() <= 1e-11 * np.abs(g["y"]).max()
        x = np.tile(g["x"], (5, 1))[:g["x"].shape[0] * 4 + 3]
        assert np.abs(plan.run(x, 2) - oracle.execute_flow(nodes, x)).max() <= 1e-11 * np.abs(g["y"]).max()
        assert plan.run(x[:0], 2).shape == (0, nodes[-1].output_dim)
    assert covered >= 1
    with pytest.raises(TypeError):
        fast_cpu.Plan(helpers.product_net(3))


@pytest.mark.parametrize("maker", [helpers.overlapping_net, helpers.linear_net, helpers.product_net])
def test_numpy_and_c_restatements_agree(maker):
    nodes = maker(3)
    x = np.random.default_rng(1).normal(size=(37, nodes[0].input_dim)) * 2
    a, b = oracle.execute_flow(nodes, x), ref_c.execute_flow(nodes, x)
    assert a.shape == (37, nodes[-1].output_dim)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()


def test_expansion_known_answers():
    x = np.array([[1.0, -2.0, 3.0], [0.0, 0.5, -4.0]])
    node = N.GeneralExpansionNode([N.identity, N.unsigned_08expo, N.signed_08expo, N.QT, N.pair_prodsadj_ex(1, "offset"),
                                   N.sel_exp(2, N.QT), N.pair_prodsadj_ex(2, "band"), N.pair_prodsadj_ex(1, "band"),
                                   N.sel_exp(2, N.pair_prodsadj_ex(2, "band"))], 3)
    y = oracle.execute_node(node, x)
    exp0 = np.array([1, -2, 3, 1, 2 ** 0.8, 3 ** 0.8, 1, -(2 ** 0.8), 3 ** 0.8,
                     1, -2, 3, 4, -6, 9, -2, -6, 1, -2, 4,
                     1, 4, 9, -2, -6,              # band of 2 offsets: squares, then x_i x_{i+1}
                     1, 4, 9,                      # band of 1 offset: the squares alone
                     1, 4, -2])                    # band of 2 over the first 2 columns
    with pytest.raises(ValueError, match="reading"):
        N.pair_prodsadj_ex(2, None)
    assert node.output_dim == exp0.size == y.shape[1]
    assert np.allclose(y[0], exp0, rtol=1e-15)
    assert y[1, 3] == 0.0 and y[1, 6] == 0.0              # 0 ** 0.8 == 0 exactly, signed too
    assert np.allclose(ref_c.execute_node(node, x), y, rtol=1e-15)


def test_rectangular_switchboard_order():
    # 4x4 single-channel image, 2x2 fields, stride 2: field-row major, then field column, then
    # row in field, column in field
    sb = N.Rectangular2dSwitchboard((4, 4), (2, 2), (2, 2), 1)
    assert sb.connections.tolist() == [0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15]
    sb2 = N.Rectangular2dSwitchboard((2, 1), (2, 1), (2, 1), 3)   # merge two 3-channel nodes along x
    assert sb2.connections.tolist() == [0, 1, 2, 3, 4, 5]
    x = np.arange(32.0).reshape(2, 16)
    assert np.array_equal(oracle.execute_node(sb, x), x[:, sb.connections])


def test_affine_nodes_definitions():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 4))
    pca = N.PCANode(rng.normal(size=4), rng.normal(size=(4, 3)))
    assert np.allclose(oracle.execute_node(pca, x), (x - pca.avg) @ pca.v)
    sfa = N.SFANode(rng.normal(size=4), rng.normal(size=(4, 2)))
    assert np.allclose(sfa._bias, sfa.avg @ sfa.sf)
    assert np.allclose(oracle.execute_node(sfa, x), (x - sfa.avg) @ sfa.sf)   # x sf - avg sf
    lr = N.LinearRegressionNode(rng.normal(size=(5, 2)))
    assert np.allclose(oracle.execute_node(lr, x), np.hstack([np.ones((5, 1)), x]) @ lr.beta)


def test_layer_and_dimension_check():
    rng = np.random.default_rng(0)
    layer = N.Layer([helpers.rand_pca(rng, 3, 2), helpers.rand_pca(rng, 5, 4)])
    x = rng.normal(size=(6, 8))
    y = oracle.execute_node(layer, x)
    assert np.allclose(y[:, :2], oracle.execute_node(layer.nodes[0], x[:, :3]))
    assert np.allclose(y[:, 2:], oracle.execute_node(layer.nodes[1], x[:, 3:]))
    with pytest.raises(ValueError):
        oracle.execute_node(layer, x[:, :7])
    assert oracle.execute_flow([layer], np.zeros((0, 8))).shape == (0, 6)      # empty batch


def test_trained_layer_is_whitened_and_slow(nets):
    """Properties the training must produce (SURVEY.md §7 'Hard parts'): PCA outputs of layer 0 are
    zero-mean / identity-covariance on the training sequence; SFA outputs are unit variance and
    ordered slowest first."""
    nodes = nets("T5L-16")
    seq = synth.make_training_sequence(600, 16)
    sb, layer = nodes[0], nodes[1]
    xin = oracle.execute_node(sb, seq)
    node0 = layer.nodes[0]
    z = oracle.execute_node(node0.flow[0], xin[:, :16])
    assert np.abs(z.mean(axis=0)).max() < 1e-9
    assert np.abs(np.cov(z.T) - np.eye(z.shape[1])).max() < 1e-8
    y = oracle.execute_node(node0, xin[:, :16])
    assert np.abs(y.var(axis=0, ddof=1) - 1).max() < 1e-8
    delta = ((y[1:] - y[:-1]) ** 2).mean(axis=0)
    assert np.all(np.diff(delta) > -1e-9)


def test_u11l_flop_count(nets):
    """11 017 088 algorithmic FLOPs per 128x128 sub-image."""
    assert synth.flops_per_row(nets("U11L-128")) == 11017088


def test_classifier_known_answers():
    """Reference-owned data: stored _sqrt_def_covs must equal det(inv_covs)^-1/2, and
    the log-domain regression equals MDP's linear-domain formula."""
    g = np.load(os.path.join(GOLD, "classifiers.npz"))
    assert int(g["n_classifiers"]) == 7          # every (classes, features) shape of /root/reference/SavedClassifiers
    assert sorted(g["c%d_means" % i].shape for i in range(7)) == [(2, 5), (10, 9), (39, 4), (39, 5), (50, 10), (50, 12), (50, 20)]
    for i in range(7):
        means, ic, sd = g["c%d_means" % i], g["c%d_inv_covs" % i], g["c%d_sqrt_def_covs" % i]
        p, avg, x = g["c%d_p" % i], g["c%d_avg_labels" % i], g["c%d_x" % i]
        sign, logdet = np.linalg.slogdet(ic)
        assert np.all(sign > 0)
        assert np.abs(np.exp(-0.5 * logdet) / sd - 1).max() < 1e-12
        reg, std = ref_c.gauss_regression(x, means, ic, sd, p, avg)
        assert np.allclose(reg, g["c%d_reg" % i], rtol=1e-12) and np.allclose(std, g["c%d_std" % i], rtol=1e-9, atol=1e-12)
        k, d = means.shape
        xm = x[:, None, :] - means[None]
        expo = -0.5 * np.einsum("nkd,kde,nke->nk", xm, ic, xm)
        prob = p * (2 * np.pi) ** (-d / 2.0) / sd * np.exp(expo - expo.max(axis=1, keepdims=True))
        prob /= prob.sum(axis=1, keepdims=True)
        assert np.allclose(prob @ avg, reg, rtol=1e-10)


@pytest.mark.parametrize("seed", range(8))
def test_fuzzed_hierarchies_host_side(seed):
    """helpers.fuzz_net: numpy and C restatements agree, the blob round-trips bit-exactly."""
    from pyfaceanalysis_amd.blob import flow_to_blob
    nodes = helpers.fuzz_net(seed)
    x = np.random.default_rng(seed).normal(size=(9, nodes[0].input_dim)) * 1.5
    a = oracle.execute_flow(nodes, x)
    assert np.abs(a - ref_c.execute_flow(nodes, x)).max() <= 1e-12 * np.abs(a).max()
    assert np.array_equal(a, oracle.execute_flow(blob_to_flow(flow_to_blob(nodes)), x))


@pytest.mark.parametrize("seed", [1, 2, 10, 22])
def test_igsfa_variants_host_side(seed):
    """iGSFA record variants (lr on scaled / unscaled features, per-column / matrix scaling): both restatements agree,
    the blob round-trips, and the rewrite the native loader applies (hg_tree.cpp normalise_igsfa: every variant
    expressed as 'per-column scale 1, lr on the scaled features') is the same map."""
    from pyfaceanalysis_amd.blob import flow_to_blob
    nodes = helpers.fuzz_igsfa_net(seed)
    x = np.random.default_rng(seed).normal(size=(11, nodes[0].input_dim)) * 1.5
    a = oracle.execute_flow(nodes, x)
    assert np.abs(a - ref_c.execute_flow(nodes, x)).max() <= 1e-12 * np.abs(a).max()
    assert np.array_equal(a, oracle.execute_flow(blob_to_flow(flow_to_blob(nodes)), x))

    def rewritten(ig):
        k = ig.sfa_node.output_dim
        M = ig.scaling_matrix if ig.scaling == "matrix" else np.diag(ig.magn_n_sfa_x.reshape(-1))
        sfa = N.SFANode(np.zeros(ig.sfa_node.input_dim), ig.sfa_node.sf @ M, ig.sfa_node._bias @ M)
        lr = ig.lr_node
        if lr is not None and ig.lr_input == "unscaled":       # lr(n) = lr(s M^-1)
            lr = N.LinearRegressionNode(np.vstack([lr.beta[0:1], np.linalg.solve(M, lr.beta[1:])]))
        return N.iGSFANode(ig.x_mean, ig.exp_node, sfa, np.ones(k), lr, ig.pca_node, ig.num_sfa_features_preserved,
                           reconstruct_with_sfa=ig.reconstruct_with_sfa)
    legacy = [N.Layer([rewritten(n) for n in nd.nodes]) if isinstance(nd, N.Layer) else nd for nd in nodes]
    b = oracle.execute_flow(legacy, x)
    assert np.abs(a - b).max() <= 1e-10 * np.abs(a).max()
