"""Synthetic hierarchical networks and synthetic sub-image batches.

The reference's trained flows (SavedNetworks/*.pckl) are absent (.MISSING_LARGE_BLOBS:1-8), so
benchmarks and tests need a stand-in with the same structure as the "Non-Linear Ultra Thin 11
Layer Network" named by every face/eye classifier file (Pipelines/Pipeline_experimental.txt:7):
11 x [Switchboard -> Layer of per-node (whitening PCA -> [x, |x|^0.8] expansion -> SFA)].
Topology and dimensions are the survey-defined "U11L-128"/"U11L-64" presets
(BASELINE.md §3); the real layer sizes are unrecoverable.

Weights are obtained by actually *training* every node (batched over the nodes of a layer with
numpy: covariance + eigendecomposition) on a seeded synthetic image
sequence, so every layer's outputs stay near zero mean / unit variance and rounding error
growth through the 11 layers is the well-conditioned kind a trained network has.  This module
is offline model construction (the counterpart of MDP ``train``/``stop_training``, which
PyFaceAnalysis itself never calls — SURVEY.md §3.4); it is not an execute path.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from . import nodes as N

INPUT_SEED = 12345600        # FaceDetectUpdated.py:146 (seed borrowed per SURVEY.md §8d)
WEIGHT_SEED = 20160623

PRESETS = {
    # name: (side, f0, [(p, s) per layer])
    "U11L-128": (128, 4, [(13, 13), (20, 20), (35, 35), (60, 60)] + [(60, 60)] * 7),
    "U11L-64": (64, 2, [(4, 8), (13, 13), (20, 20), (35, 35)] + [(60, 60)] * 7),
    # small nets for tests / golden fixtures
    "T5L-16": (16, 4, [(10, 9), (12, 10), (14, 12), (16, 12), (18, 10)]),
    "T3L-8": (8, 4, [(9, 7), (10, 8), (12, 6)]),
}


def _box3(img):
    """3x3 box filter with edge replication over the last two axes (float64)."""
    p = np.pad(img.astype(np.float64), [(0, 0)] * (img.ndim - 2) + [(1, 1), (1, 1)], mode="edge")
    h, w = img.shape[-2], img.shape[-1]
    out = np.zeros(img.shape, dtype=np.float64)
    for dy in range(3):
        for dx in range(3):
            out += p[..., dy:dy + h, dx:dx + w]
    return out / 9.0


def make_subimages(n, side=128, seed=INPUT_SEED, dtype=np.float32, chunk=1024):
    """(n, side*side) batch shaped like ``images_asarray`` output (face_analysis.py:786):
    row-major pixels, integer values 0..255.  uint8 noise, 3x3 box-filtered, rounded
    (SURVEY.md §8d "Synthetic inputs")."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, side * side), dtype=dtype)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        raw = rng.integers(0, 256, (m, side, side), dtype=np.uint8)
        out[i0:i0 + m] = np.rint(_box3(raw)).reshape(m, side * side)
    return out


def make_training_sequence(n, side, seed=WEIGHT_SEED):
    """Slowly varying image sequence: a side x side window gliding (<= 1 px / step) over a
    box-filtered noise texture, so that consecutive frames are correlated (SFA needs a slow
    variable) while single frames have the statistics of ``make_subimages``."""
    rng = np.random.default_rng(seed)
    ext = 3 * side
    tex = np.rint(_box3(rng.integers(0, 256, (ext + side, ext + side), dtype=np.uint8)))
    t = np.arange(n, dtype=np.float64)
    # smooth Lissajous path with speed < 1 px/step
    fx, fy = 0.9 / ext * 2.0, 0.9 / ext * 2.0 * 0.731
    px = np.rint((0.5 + 0.5 * np.sin(fx * t + rng.uniform(0, 6.28))) * (ext - 1)).astype(int)
    py = np.rint((0.5 + 0.5 * np.sin(fy * t + rng.uniform(0, 6.28))) * (ext - 1)).astype(int)
    seq = np.empty((n, side * side), dtype=np.float64)
    for i in range(n):
        seq[i] = tex[py[i]:py[i] + side, px[i]:px[i] + side].reshape(-1)
    # slow global illumination drift keeps the DC component from being trivially constant
    gain = 1.0 + 0.08 * np.sin(2 * np.pi * t / 977.0)
    seq = np.clip(np.rint((seq - 127.5) * gain[:, None] + 127.5), 0, 255)
    return seq


def _sign_fix(v):
    """Deterministic eigenvector signs: largest-|.| entry of every column positive. v: (..., d, k)"""
    idx = np.argmax(np.abs(v), axis=-2)
    s = np.sign(np.take_along_axis(v, idx[..., None, :], axis=-2))
    s[s == 0] = 1.0
    return v * s


def _gram(a, b=None):
    """a, b: (n, T, d) / (n, T, e) -> (n, d, e) = a^T b per node (batched BLAS matmul)."""
    b = a if b is None else b
    return np.matmul(np.swapaxes(a, 1, 2), b)


def _batched_cov(x):
    """x: (n, T, d) node-major -> mean (n, d), covariance (n, d, d)."""
    mu = x.mean(axis=1)
    xc = x - mu[:, None, :]
    return mu, _gram(xc) / (x.shape[1] - 1)


def _expand(z, expo):
    return np.concatenate([z, np.abs(z) ** expo], axis=-1)


def _train_pca(x, p, whiten=True, floor=1e-9):
    mu, cov = _batched_cov(x)
    lam, vec = np.linalg.eigh(cov)                      # ascending
    lam, vec = lam[:, ::-1][:, :p], vec[:, :, ::-1][:, :, :p]
    vec = _sign_fix(vec)
    if whiten:
        lam = np.maximum(lam, floor * lam[:, :1])
        vec = vec / np.sqrt(lam)[:, None, :]
    return mu, vec


def _train_sfa(e, s):
    """Generalised eigenproblem  Cov(de) w = lambda Cov(e) w, slowest first, w' Cov(e) w = 1."""
    mu, B = _batched_cov(e)
    de = e[:, 1:] - e[:, :-1]
    A = _gram(de) / (de.shape[1] - 1)
    lamB, VB = np.linalg.eigh(B)
    lamB = np.maximum(lamB, 1e-10 * lamB[:, -1:])
    S = VB / np.sqrt(lamB)[:, None, :]                  # whitening of e
    Aw = np.matmul(np.swapaxes(S, 1, 2), np.matmul(A, S))
    Aw = 0.5 * (Aw + np.swapaxes(Aw, 1, 2))
    lam, U = np.linalg.eigh(Aw)                         # ascending = slowest first
    W = np.matmul(S, U[:, :, :s])
    return mu, _sign_fix(W), lam[:, :s]


def train_hierarchy(side, f0, layer_dims, n_train=1500, seed=WEIGHT_SEED, node_kind="pca_exp_sfa",
                    expo=0.8, layout="flownode", verbose=False, device=None):
    """Build + train the hierarchy.  Returns the list of top-level nodes
    ``[Switchboard, Layer, Switchboard, Layer, ...]`` (a ``Flow.flow`` list).

    node_kind : "pca_exp_sfa" (whitening PCA -> [x,|x|^expo] -> SFA)
                "igsfa"       (iGSFANode per node; row a8)
    layout    : "flownode"  -> Layer([FlowNode([PCA, Exp, SFA]), ...])
                "separate"  -> Layer([PCA...]), Layer([Exp...]), Layer([SFA...])  (how
                               cuicuilco's network builder stacks them)
    device    : None -> numpy on the host (below); a HIP device ordinal -> statistics, eigen-solves and the
                layer-to-layer passes on that GPU (pyfaceanalysis_amd.train.train_hierarchy_device, "pca_exp_sfa" only)
    """
    if device is not None:
        if node_kind != "pca_exp_sfa":
            raise ValueError("device training covers node_kind 'pca_exp_sfa'")
        from .train import train_hierarchy_device
        return train_hierarchy_device(side, f0, layer_dims, n_train=n_train, seed=seed, expo=expo, layout=layout, device=device,
                                      verbose=verbose)
    x = make_training_sequence(n_train, side, seed)                    # (T, side*side)
    grid = None
    flow = []
    for li, (p, s) in enumerate(layer_dims):
        if li == 0:
            sb = N.Rectangular2dSwitchboard((side, side), (f0, f0), (f0, f0), 1)
        else:
            nx, ny = grid
            merge_x = (li % 2 == 1)
            if merge_x and nx == 1:
                merge_x = False
            if not merge_x and ny == 1:
                merge_x = True
            field = (2, 1) if merge_x else (1, 2)
            sb = N.Rectangular2dSwitchboard((nx, ny), field, field, ch)
        grid = sb.out_channels_xy
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        xin = np.ascontiguousarray(                                     # (n, T, d_in) node-major
            x[:, sb.connections].reshape(x.shape[0], n_nodes, d_in).transpose(1, 0, 2))
        if node_kind == "pca_exp_sfa":
            p_ = min(p, d_in)
            s_ = min(s, 2 * p_)
            mu, v = _train_pca(xin, p_, whiten=True)
            z = np.matmul(xin - mu[:, None, :], v)
            e = _expand(z, expo)
            mue, sf, _lam = _train_sfa(e, s_)
            y = np.matmul(e - mue[:, None, :], sf)
            funcs = [N.identity, N.unsigned_expo(expo) if expo != 0.8 else N.unsigned_08expo]
            pcas = [N.WhiteningNode(mu[k], v[k]) for k in range(n_nodes)]
            exps = [N.GeneralExpansionNode(funcs, p_) for _ in range(n_nodes)]
            sfas = [N.SFANode(mue[k], sf[k]) for k in range(n_nodes)]
            if layout == "flownode":
                layer = [N.Layer([N.FlowNode([pcas[k], exps[k], sfas[k]]) for k in range(n_nodes)])]
            else:
                layer = [N.Layer(pcas), N.Layer(exps), N.Layer(sfas)]
            out_dim = s_
        elif node_kind == "igsfa":
            out_dim = min(s, d_in)
            k_sfa = max(1, min(out_dim - 1, out_dim // 2)) if out_dim > 1 else 1
            mu = xin.mean(axis=1)
            x0 = xin - mu[:, None, :]
            scale0 = np.sqrt((x0 ** 2).mean(axis=(1, 2)))[:, None, None] + 1e-12
            e = _expand(x0 / scale0, expo)                # train on scale-normalised data ...
            mue, sf, _lam = _train_sfa(e, k_sfa)
            sv = np.matmul(e - mue[:, None, :], sf)
            # ... but the node expands x0 itself: fold the normalisation into sf/avg
            sc = np.concatenate([np.broadcast_to(1.0 / scale0[:, 0], (n_nodes, d_in)),
                                 np.broadcast_to(1.0 / scale0[:, 0] ** expo, (n_nodes, d_in))], axis=1)
            sf_n = sf * sc[:, :, None]
            mue_n = mue / sc
            # least-squares reconstruction of x0 from s (no intercept needed: both zero-mean)
            G = _gram(sv)
            H = _gram(sv, x0)
            beta = np.linalg.solve(G, H)                                    # (n, k, d_in)
            magn = np.sqrt((beta ** 2).sum(axis=2)) + 1e-12                # (n, k)
            beta_n = beta / magn[:, :, None]
            nsv = sv * magn[:, None, :]
            r = x0 - np.matmul(nsv, beta_n)
            q_dim = out_dim - k_sfa
            if q_dim > 0:
                mur, vr = _train_pca(r, q_dim, whiten=False)
                q = np.matmul(r - mur[:, None, :], vr)
            else:
                raise ValueError("igsfa preset needs out_dim >= 2")
            y = np.concatenate([nsv, q], axis=2)
            funcs = [N.identity, N.unsigned_expo(expo) if expo != 0.8 else N.unsigned_08expo]
            nodes = []
            for k in range(n_nodes):
                lr = N.LinearRegressionNode(np.vstack([np.zeros((1, d_in)), beta_n[k]]))
                nodes.append(N.iGSFANode(mu[k], N.GeneralExpansionNode(funcs, d_in),
                                         N.SFANode(mue_n[k], sf_n[k]), magn[k], lr,
                                         N.PCANode(mur[k], vr[k]), k_sfa))
            layer = [N.Layer(nodes)]
        else:
            raise ValueError("unknown node_kind %r" % (node_kind,))
        flow.append(sb)
        flow.extend(layer)
        x = np.ascontiguousarray(y.transpose(1, 0, 2)).reshape(y.shape[1], n_nodes * out_dim)
        ch = out_dim
        if verbose:
            print("  L%-2d grid %-7s nodes %4d  d_in %3d -> %3d   out std %.3f"
                  % (li, grid, n_nodes, d_in, out_dim, float(x.std())))
        if n_nodes == 1 and li < len(layer_dims) - 1:
            break
    return flow


def build_preset(name="U11L-128", n_train=None, seed=WEIGHT_SEED, node_kind="pca_exp_sfa",
                 layout="flownode", verbose=False, device=None):
    side, f0, dims = PRESETS[name]
    if n_train is None:
        n_train = 1500 if side >= 64 else 600
    return train_hierarchy(side, f0, dims, n_train=n_train, seed=seed, node_kind=node_kind,
                           layout=layout, verbose=verbose, device=device)


def preset_input_side(name):
    return PRESETS[name][0]


def cached_preset_blob(name="U11L-128", cache_dir=None, **kw):
    """Blob of a preset, cached on disk (training U11L-128 takes ~1 min of numpy)."""
    from .blob import flow_to_blob, blob_to_flow
    key = hashlib.sha1(repr((name, sorted(kw.items()), WEIGHT_SEED, 3)).encode()).hexdigest()[:12]
    cache_dir = cache_dir or os.environ.get("HIGSFA_CACHE",
                                            os.path.join(os.path.expanduser("~"), ".cache", "higsfa"))
    path = os.path.join(cache_dir, "%s-%s.hgflow" % (name, key))
    if os.path.exists(path):
        with open(path, "rb") as f:
            blob = f.read()
        return blob, blob_to_flow(blob)
    flow = build_preset(name, **kw)
    blob = flow_to_blob(flow)
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
    except OSError:
        pass
    return blob, flow


def flops_per_row(flow_nodes):
    """Algorithmic FLOPs per sub-image: 2*in*out per affine node (SURVEY.md §8d formula)."""
    total = 0
    for n in flow_nodes:
        if isinstance(n, N.Layer):
            total += flops_per_row(n.nodes)
        elif isinstance(n, N.FlowNode):
            total += flops_per_row(n.flow)
        elif isinstance(n, (N.PCANode, N.SFANode, N.LinearRegressionNode)):
            total += 2 * n.input_dim * n.output_dim
        elif isinstance(n, N.iGSFANode):
            total += flops_per_row([x for x in (n.sfa_node, n.lr_node, n.pca_node) if x is not None])
    return total
