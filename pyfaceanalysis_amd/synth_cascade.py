"""Synthetic stand-in for the reference's detection pipeline (Pipelines/Pipeline_experimental.txt): the 17 face stages
(FaceDetectUpdated.py:665 ``range(num_networks - 5)``) with the pipeline's structure — which stages own a network, which
reuse the previous features (``None0``), classifier input widths 9 / 10 / 20 — but synthetic networks (the trained flows
are stripped, .MISSING_LARGE_BLOBS) and synthetic Gaussian classifiers calibrated on the synthetic networks' features.
It exists so that BASELINE.json configs[2] (one 1920x1080 frame through the whole pyramid and cascade) can be run and
timed; it detects nothing meaningful.  Offline construction, not an execute path.
"""
from __future__ import annotations

import numpy as np

from . import grid
from .cascade import CUT_OFFS_FACE, Stage
from .classifier import GaussianClassifier

# (stage name, owns a network, classifier input width) — Pipelines/Pipeline_experimental.txt:5-55
FACE_STAGES = [("Disc1", True, 9), ("PosX0", True, 10), ("PosY0", False, 10), ("PAng0", False, 20), ("Scale0", False, 20),
               ("Disc3", True, 9), ("PosX1", True, 20), ("PosY1", False, 20), ("PAng1", False, 20), ("Scale1", False, 20),
               ("Disc5", True, 9), ("PosX2", True, 20), ("PosY2", False, 20), ("PAng2", False, 20), ("Scale2", False, 20),
               ("Disc7", True, 9), ("Disc9", True, 9)]


def _soft_regression(means, inv_cov, p, labels, x):
    """Posterior-weighted labels of a shared-covariance Gaussian classifier, numpy float64 (calibration of the synthetic
    classifiers only — offline construction; the execute path's regression is hg_gauss_regression)."""
    dm = x[:, None, :] - means[None]
    e = -0.5 * np.einsum("nkd,de,nke->nk", dm, inv_cov, dm) + np.log(p)[None]
    w = np.exp(e - e.max(axis=1, keepdims=True))
    return (w / w.sum(axis=1, keepdims=True)) @ labels


def quantile_classifier(feats, d, labels, ridge=1e-3, device=0, pass_fraction=None, cut_off=None, descending=False):
    """K = len(labels) Gaussian classes along the quantiles of the first feature: class means from the sample, one pooled
    covariance, priors = bin fractions, avg_labels = labels.  Parameters in the layout of the reference's classifier
    pickles (means, inv_covs, _sqrt_def_covs, p, avg_labels — SURVEY.md §8f-2).
    ``pass_fraction`` / ``cut_off``: scale the labels (regression is linear in them) so that this share of the calibration
    sample regresses below ``cut_off`` — a Disc stage that lets about that share of a like population through."""
    f = np.asarray(feats, dtype=np.float64)[:, :d]
    k = len(labels)
    order = np.argsort(-f[:, 0] if descending else f[:, 0], kind="stable")
    bins = np.array_split(order, k)
    means = np.stack([f[b].mean(axis=0) for b in bins])
    resid = np.concatenate([f[b] - means[i] for i, b in enumerate(bins)])
    cov = resid.T @ resid / max(len(resid) - k, 1) + ridge * np.eye(d) * max(np.trace(resid.T @ resid) / len(resid) / d, 1e-12)
    inv = np.linalg.inv(cov)
    sqrt_det = np.sqrt(np.linalg.det(cov))
    p = np.array([len(b) for b in bins], dtype=np.float64) / len(f)
    labels = np.asarray(labels, dtype=np.float64)
    if pass_fraction is not None:
        r = _soft_regression(means, inv, p, labels, f)
        t = max(float(np.quantile(r, pass_fraction)), 1e-12)
        labels = labels * (cut_off / t)
    return GaussianClassifier(means, np.stack([inv] * k), np.full(k, sqrt_det), p, labels=np.arange(k),
                              avg_labels=np.asarray(labels, dtype=np.float64), device=device)


# which of the pipeline's trained flows a network-owning stage runs (Pipelines/Pipeline_experimental.txt):
# FaceCentering2 _1468510885 for Disc1/3/5/7, _1470325647 for Disc9, RTransXYPAngScale _1468325214 for iteration 0 and
# _1468332771 for iterations 1-2 — four flows of one architecture
FLOW_ROLE = {"Disc1": 0, "Disc3": 0, "Disc5": 0, "Disc7": 0, "Disc9": 1, "PosX0": 2, "PosX1": 3, "PosX2": 3}
# classes of the reference's classifier files (SavedClassifiers/*.pckl): Disc (10, 9); pose regressors (50, 10) / (50, 20)
N_CLASSES = {"Disc": 10, "PosX": 50, "PosY": 50, "PAng": 50, "Scale": 50}


def build_face_cascade(flow, features, pipeline=None, keep_fraction=0.1, n_classes=None, device=0, stages=None, later_keep_fraction=None):
    """Stages of the synthetic face cascade.

    ``flow``: ONE Flow that every network-owning stage runs, or a list of four (the reference's pipeline uses four trained
    flows of one architecture: ``FLOW_ROLE``).  ``features``: (m, >= 20) features of a sample of windows — one array, or one
    per flow — used to calibrate the classifiers so that the first Disc stage keeps about ``keep_fraction`` of the windows
    and the pose stages propose small corrections inside their training ranges (Pipeline header: Dx 40, Dy 20, Dang 22.5,
    scale 0.694..0.981).  ``n_classes``: classes of every classifier (default: the reference's files' — 10 for Disc, 50 for
    the pose regressors).  ``stages``: another (name, owns a network, classifier width) list than ``FACE_STAGES``.
    ``later_keep_fraction``: target pass rate of every Disc stage after the first (default: keep_fraction).  Each Disc
    classifier's labels are scaled so that the intended share of the calibration sample regresses below the stage's own
    cut-off (cut_offs_face, FaceDetectUpdated.py:98) — cumulative for stages that share a flow, whose scores are
    correlated — which thins the candidates gradually (1738 -> ~350 -> ~140 -> ~55 -> ~20 -> a few on the 1080p frame)."""
    p = dict(grid.FACE_PIPELINE if pipeline is None else pipeline)
    flows = list(flow) if isinstance(flow, (list, tuple)) else [flow] * 4
    feats = list(features) if isinstance(features, (list, tuple)) else [features] * 4
    if len(flows) != 4 or len(feats) != 4:
        raise ValueError("build_face_cascade: one flow or four (FLOW_ROLE), with one feature sample each")

    def labels(kind, k):
        return {
            "Disc": (np.arange(k) + 0.5) / k,          # rising with the class's quantile position; scaled per stage below
            # small corrections: the synthetic networks carry no face semantics, so a window that moved far would get unrelated
            # features at the next Disc stage and the cascade would die out after two iterations instead of exercising all 17
            "PosX": np.linspace(-0.03 * p["net_Dx"], 0.03 * p["net_Dx"], k),
            "PosY": np.linspace(-0.03 * p["net_Dy"], 0.03 * p["net_Dy"], k),
            "PAng": np.linspace(-0.1 * p["net_Dang"], 0.1 * p["net_Dang"], k),
            "Scale": np.linspace(0.815, 0.835, k),
        }[kind]
    out = []
    role = 0
    passed = {}            # per flow: share of a random window population that the Disc stages so far (on that flow) let through
    for name, own, d in (FACE_STAGES if stages is None else stages):
        if own:
            role = FLOW_ROLE.get(name, 0)          # a stage without a network reads the features of the last flow that ran
        f = np.asarray(feats[role])
        d = min(d, f.shape[1])                     # small test networks have fewer than 20 outputs
        k = N_CLASSES[name[:-1]] if n_classes is None else n_classes
        k = max(2, min(k, len(f) // 4))            # tiny calibration samples: fewer classes than rows
        kw = {}
        if name[:-1] == "Disc":
            # Stages that share a flow see correlated scores (the reference runs Disc1/3/5/7 on ONE flow): each thins what the
            # previous one on that flow let through, so its target is the cumulative share; an unrelated flow starts afresh.
            share = keep_fraction if not passed else (keep_fraction if later_keep_fraction is None else later_keep_fraction)
            passed[id(flows[role])] = passed.get(id(flows[role]), 1.0) * share
            # synthetic networks trained on like data learn the same slowest feature up to its sign: classes of another flow are
            # ordered the way that agrees with the first flow's, so that "face-like" means the same end of the feature for both
            flip = bool(role != 0 and len(feats[role]) == len(feats[0]) and
                        np.corrcoef(np.asarray(feats[role])[:, 0], np.asarray(feats[0])[:, 0])[0, 1] < 0)
            kw = dict(pass_fraction=passed[id(flows[role])], cut_off=CUT_OFFS_FACE[int(name[-1])], descending=flip)
        out.append(Stage(name, flows[role] if own else None, quantile_classifier(f, d, labels(name[:-1], k), device=device, **kw)))
    return out
