"""Shared test helpers: random (untrained) flows with awkward structure."""
import numpy as np

from pyfaceanalysis_amd import nodes as N


def rand_pca(rng, d_in, d_out, cls=N.PCANode):
    return cls(rng.normal(size=d_in), rng.normal(size=(d_in, d_out)) / np.sqrt(d_in))


def rand_sfa(rng, d_in, d_out, cls=N.SFANode):
    return cls(rng.normal(size=d_in) * 0.1, rng.normal(size=(d_in, d_out)) / np.sqrt(d_in))


def overlapping_net(seed=0, funcs=None):
    """8x8 input, 4x4 fields with stride 2 (overlap), then a 3-node merge with overlap; exercises
    switchboards that reuse inputs, uneven node sizes and sel_exp."""
    rng = np.random.default_rng(seed)
    funcs = funcs or [N.identity, N.unsigned_08expo, N.sel_exp(3, N.signed_08expo)]
    sb0 = N.Rectangular2dSwitchboard((8, 8), (4, 4), (2, 2), 1)          # 3x3 = 9 nodes of 16 inputs
    l0 = []
    for k in range(9):
        p = 5 + (k % 3)                                                    # uneven PCA widths 5..7
        ex = N.GeneralExpansionNode(funcs, p)
        l0.append(N.FlowNode([rand_pca(rng, 16, p, N.WhiteningNode), ex, rand_sfa(rng, ex.output_dim, 6)]))
    sb1 = N.Rectangular2dSwitchboard((3, 3), (2, 2), (1, 1), 6)           # 2x2 = 4 nodes of 24 inputs, overlapping
    l1 = []
    for k in range(4):
        ex = N.GeneralExpansionNode(funcs, 9)
        l1.append(N.FlowNode([rand_pca(rng, 24, 9), ex, rand_sfa(rng, ex.output_dim, 18 + k, N.GSFANode)]))
    rng2 = np.random.default_rng(seed + 1)
    conn = rng2.permutation(sum(18 + k for k in range(4)))[:40]            # irregular switchboard: permuted subset
    sb2 = N.PInvSwitchboard(sum(18 + k for k in range(4)), conn)
    ex = N.GeneralExpansionNode(funcs, 12)
    top = N.Layer([N.FlowNode([rand_pca(rng, 40, 12), ex, rand_sfa(rng, ex.output_dim, 20)])])
    return [sb0, N.Layer(l0), sb1, N.Layer(l1), sb2, top]


def linear_net(seed=0):
    """Linear PCA network (the reference's age net is 'linearPCANetworkU11L'):
    Pipelines/Pipeline_experimental.txt:64) with a CloneLayer and consecutive affines that fold."""
    rng = np.random.default_rng(seed)
    sb0 = N.Rectangular2dSwitchboard((8, 8), (4, 4), (4, 4), 1)
    l0 = N.CloneLayer(rand_pca(rng, 16, 7), 4)
    sb1 = N.Rectangular2dSwitchboard((2, 2), (2, 2), (2, 2), 7)
    l1 = N.Layer([N.FlowNode([rand_pca(rng, 28, 12), rand_sfa(rng, 12, 9)])])
    return [sb0, l0, sb1, l1]


def linear_u11l_96(seed=0):
    """The shape of the reference's age network: 'linearPCANetworkU11L' fed 96x96 sub-images
    (Pipelines/Pipeline_experimental.txt:4,64; call site face_analysis.py:1257): 96 = 3 * 32, so 3x3 fields give 32x32
    first-layer nodes and ten pair merges reach one node — 11 layers, every node linear (PCA then SFA, which fold into one
    affine map).  Random weights (the trained flow is not shipped); a side that is not 4 * 2^k and 9-pixel fields."""
    rng = np.random.default_rng(seed)
    dims = [6, 9, 12, 16, 20, 24, 28, 32, 36, 40, 30]
    flow, c, w, h = [], 1, 96, 96
    for li, s_out in enumerate(dims):
        field = (3, 3) if li == 0 else ((2, 1) if li % 2 == 1 else (1, 2))
        sb = N.Rectangular2dSwitchboard((w, h), field, field, c)
        d_in = sb.out_channel_dim
        p = min(d_in, s_out + 2)
        layer = N.Layer([N.FlowNode([rand_pca(rng, d_in, p), rand_sfa(rng, p, s_out)]) for _ in range(sb.output_channels)])
        flow += [sb, layer]
        w, h = sb.out_channels_xy
        c = s_out
    assert (w, h) == (1, 1) and len(flow) == 22
    return flow


def product_net(seed=0):
    """Expansions with cross-column products (QT, pair products), HeadNode, CutoffNode: generic plan."""
    rng = np.random.default_rng(seed)
    funcs = [N.identity, N.QT, N.pair_prodsadj_ex(1, "offset"), N.sel_exp(4, N.pair_prodsadj_ex(2, "band"))]
    sb0 = N.Rectangular2dSwitchboard((8, 8), (4, 4), (4, 4), 1)
    l0 = []
    for _ in range(4):
        ex = N.GeneralExpansionNode(funcs, 5)
        l0.append(N.FlowNode([rand_pca(rng, 16, 5), ex, N.CutoffNode(ex.output_dim, -3.0, 3.0),
                              rand_sfa(rng, ex.output_dim, 8), N.HeadNode(8, 6)]))
    return [sb0, N.Layer(l0), N.IdentityNode(24), rand_pca(rng, 24, 10)]


def product_hier_net(seed=0, side=64):
    """A hierarchy of realistic size with product expansions in every layer: side x side input, 4x4 fields (256 nodes at
    side 64), then pair merges down to one node; node = whitening PCA -> [x, QT of the first 6, adjacent pair products] ->
    SFA.  Used to compare the fused and the generic plan at BASELINE batch sizes."""
    rng = np.random.default_rng(seed)
    dims = [(10, 12), (14, 16), (16, 20), (20, 24), (24, 24), (24, 24), (24, 24), (24, 20)]
    flow, c, w, h = [], 1, side, side
    for li, (p, s_out) in enumerate(dims):
        field = (4, 4) if li == 0 else ((2, 1) if (li % 2 == 1 and w > 1) or h == 1 else (1, 2))
        sb = N.Rectangular2dSwitchboard((w, h), field, field, c)
        funcs = [N.identity, N.sel_exp(6, N.QT), N.pair_prodsadj_ex(1, "offset"), N.unsigned_08expo]
        nodes = []
        for _ in range(sb.output_channels):
            ex = N.GeneralExpansionNode(funcs, p)
            nodes.append(N.FlowNode([rand_pca(rng, sb.out_channel_dim, p, N.WhiteningNode), ex, N.CutoffNode(ex.output_dim, -4.0, 4.0),
                                     rand_sfa(rng, ex.output_dim, s_out)]))      # the clip keeps the untrained products bounded
        flow += [sb, N.Layer(nodes)]
        w, h = sb.out_channels_xy
        c = s_out
        if w * h == 1:
            break
    return flow


def fuzz_product_net(seed):
    """Random hierarchy whose expansions contain cross-column products (QT, pair_prodsadj*, sel_exp of them) next to
    element-wise functions, uneven node widths, optional CutoffNode after the expansion and HeadNode after the node."""
    rng = np.random.default_rng(7000 + seed)
    w, h = (int(v) for v in rng.choice([4, 6, 8], 2))
    flow, c = [], 1
    for depth in range(int(rng.integers(1, 4))):
        fx, fy = ((int(v) for v in rng.choice([2, 3, 4], 2)) if depth == 0 else [(2, 1), (1, 2), (2, 2)][int(rng.integers(0, 3))])
        fx, fy = min(fx, w), min(fy, h)
        while w % fx:
            fx -= 1
        while h % fy:
            fy -= 1
        sb = N.Rectangular2dSwitchboard((w, h), (fx, fy), (fx, fy), c)
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        p0 = int(rng.integers(3, min(d_in, 12) + 1))
        s0 = int(rng.integers(2, 30))
        # both readings of pair_prodsadj{k}_ex (nodes.pair_prodsadj_ex): x_i x_{i+k} only ("offset"), offsets 0 .. k-1 ("band")
        pool = [N.identity, N.QT, N.pair_prodsadj_ex(1, "offset"), N.pair_prodsadj_ex(2, "offset"), N.unsigned_08expo,
                N.signed_expo(float(rng.uniform(0.6, 1.2))), N.sel_exp(int(rng.integers(2, p0)), N.QT),
                N.sel_exp(int(rng.integers(3, p0 + 1)), N.pair_prodsadj_ex(1, "offset")), N.pair_prodsadj_ex(2, "band"),
                N.pair_prodsadj_ex(1, "band"), N.sel_exp(int(rng.integers(3, p0 + 1)), N.pair_prodsadj_ex(3, "band"))]
        nf = int(rng.integers(2, 6))
        funcs = [pool[int(i)] for i in rng.choice(len(pool), nf, replace=False)]
        if not any(f.kind in ("quadratic", "pair_adj", "pair_band") for f in funcs):
            funcs[-1] = N.QT
        uneven = rng.random() < 0.4
        clip = rng.random() < 0.4
        lo, hi = (-float(rng.uniform(1.5, 4.0)), float(rng.uniform(1.5, 4.0)))

        def make(k):
            p = max(3, p0 - (k % 2)) if uneven else p0
            ex = N.GeneralExpansionNode(funcs, p)
            seq = [rand_pca(rng, d_in, p, N.WhiteningNode if rng.random() < 0.5 else N.PCANode), ex]
            if clip:
                seq.append(N.CutoffNode(ex.output_dim, lo, hi))
            seq.append(rand_sfa(rng, ex.output_dim, s0 + 2))
            seq.append(N.HeadNode(s0 + 2, s0))
            return N.FlowNode(seq)

        flow += [sb, N.Layer([make(k) for k in range(n_nodes)])]
        w, h = sb.out_channels_xy
        c = s0
        if w * h == 1:
            break
    return flow


def wide_merge_net(seed=0):
    """A layer whose nodes merge 40 children of 13 features each (520 inputs -> 40 K-blocks x 4 tiles of
    weights = 160+ KiB per node): beyond what a workgroup's LDS holds, so the loader must pick the generic plan."""
    rng = np.random.default_rng(seed)
    l0 = N.Layer([N.FlowNode([rand_pca(rng, 4, 13), N.GeneralExpansionNode([N.identity], 13), rand_sfa(rng, 13, 13)])
                  for _ in range(40)])
    ex = N.GeneralExpansionNode([N.identity, N.unsigned_08expo], 50)
    l1 = N.Layer([N.FlowNode([rand_pca(rng, 520, 50), ex, rand_sfa(rng, 100, 40)])])
    return [l0, l1]


def fuzz_net(seed):
    """Random hierarchy for planner / kernel fuzzing: random grid, field sizes, overlaps, merge directions,
    uneven per-node widths, random element-wise expansions (incl. sel_exp and odd exponents), optional
    linear layers and clone layers.  Everything the fused plan claims to cover, nothing it does not."""
    rng = np.random.default_rng(1000 + seed)
    w, h = (int(v) for v in rng.choice([6, 8, 12, 16], 2))
    pool = [N.identity, N.unsigned_08expo, N.signed_08expo, N.unsigned_expo(float(rng.uniform(0.5, 1.5))),
            N.signed_expo(float(rng.uniform(0.6, 1.2)))]
    flow, c = [], 1
    for depth in range(int(rng.integers(1, 5))):
        if depth == 0:
            fx, fy = (int(v) for v in rng.choice([2, 3, 4], 2))
        else:
            fx, fy = [(2, 1), (1, 2), (2, 2)][int(rng.integers(0, 3))]
        fx, fy = min(fx, w), min(fy, h)
        sx = fx if rng.random() < 0.7 or fx == 1 else fx - 1
        sy = fy if rng.random() < 0.7 or fy == 1 else fy - 1
        while (w - fx) % sx:
            sx -= 1
        while (h - fy) % sy:
            sy -= 1
        sb = N.Rectangular2dSwitchboard((w, h), (fx, fy), (sx, sy), c)
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        uneven = rng.random() < 0.4
        clone = not uneven and rng.random() < 0.2
        linear = rng.random() < 0.15
        p0, s0 = int(rng.integers(2, min(d_in, 40) + 1)), int(rng.integers(2, 50))
        nf = int(rng.integers(1, 4))
        funcs = [pool[int(i)] for i in rng.choice(len(pool), nf, replace=False)]
        if rng.random() < 0.3 and p0 > 4:      # one expansion per layer, like the reference's networks
            funcs[-1] = N.sel_exp(int(rng.integers(1, p0 - 2)), funcs[-1])

        def make(k):
            p = max(2, p0 - (k % 3)) if uneven else p0
            if linear:
                return N.FlowNode([rand_pca(rng, d_in, p), rand_sfa(rng, p, s0)])
            ex = N.GeneralExpansionNode(funcs, p)
            return N.FlowNode([rand_pca(rng, d_in, p, N.WhiteningNode if rng.random() < 0.5 else N.PCANode), ex,
                               rand_sfa(rng, ex.output_dim, s0, N.GSFANode if rng.random() < 0.5 else N.SFANode)])

        layer = N.CloneLayer(make(0), n_nodes) if clone else N.Layer([make(k) for k in range(n_nodes)])
        if not clone and len({n.output_dim for n in layer.nodes}) > 1:
            # the next rectangular switchboard needs one channel width: even the layer out with a head per node
            m = min(n.output_dim for n in layer.nodes)
            layer = N.Layer([N.FlowNode(list(n.flow) + [N.HeadNode(n.output_dim, m)]) if n.output_dim > m else n
                             for n in layer.nodes])
        flow += [sb, layer]
        w, h = sb.out_channels_xy
        c = layer.output_dim // n_nodes
        if w * h == 1:
            break
    return flow


def fuzz_igsfa_net(seed, lr_input=None, scaling=None):
    """Random hierarchy of iGSFA nodes: random widths, with / without the linear
    reconstruction and the expansion, merges in x / y / 2x2, uneven slow-feature counts across layers.
    The node variants (which slow features the reconstruction reads; per-column or matrix scaling) cycle with
    the seed unless given."""
    rng = np.random.default_rng(5000 + seed)
    rng_v = np.random.default_rng(9000 + seed)       # separate stream: the nets of round 1 keep their weights
    lr_input = lr_input or ("unscaled" if seed % 3 == 1 else "scaled")
    scaling = scaling or ("matrix" if seed % 4 == 2 else "per_column")
    w, h = (int(v) for v in rng.choice([4, 6, 8], 2))
    fx, fy = (int(v) for v in rng.choice([2, 3, 4], 2))
    fx, fy = min(fx, w), min(fy, h)
    while w % fx:
        fx -= 1
    while h % fy:
        fy -= 1
    flow, c = [], 1
    for depth in range(int(rng.integers(1, 4))):
        if depth:
            fx, fy = [(2, 1), (1, 2), (2, 2)][int(rng.integers(0, 3))]
            fx, fy = min(fx, w), min(fy, h)
            while w % fx:
                fx -= 1
            while h % fy:
                fy -= 1
        sb = N.Rectangular2dSwitchboard((w, h), (fx, fy), (fx, fy), c)
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        if d_in > 128:
            break
        k = int(rng.integers(1, min(d_in, 30) + 1))           # slow features (all preserved)
        q = int(rng.integers(1, min(d_in, 64 - k) + 1))      # pca part
        with_lr = rng.random() < 0.8
        funcs = [N.identity, N.unsigned_08expo] if rng.random() < 0.7 else [N.identity]
        if rng.random() < 0.2:
            funcs = [N.identity, N.signed_expo(float(rng.uniform(0.6, 1.1)))]

        def make():
            ex = N.GeneralExpansionNode(funcs, d_in)
            sfa = rand_sfa(rng, ex.output_dim, k, N.GSFANode)
            lr = N.LinearRegressionNode(rng.normal(size=(k + 1, d_in)) / np.sqrt(k + 1))
            pca = rand_pca(rng, d_in, q)
            pca.avg = np.zeros_like(pca.avg)                   # the residual is centred by construction
            magn = rng.uniform(0.5, 2.0, size=k)
            mat = None
            if scaling == "matrix":        # like the R of a QR decomposition: triangular, well conditioned
                mat = np.triu(rng_v.normal(size=(k, k)) * 0.3, 1) + np.diag(rng_v.uniform(0.5, 2.0, size=k))
            return N.iGSFANode(rng.normal(size=d_in), ex, sfa, None if mat is not None else magn, lr, pca, k,
                               reconstruct_with_sfa=with_lr, lr_input=lr_input, scaling=scaling, scaling_matrix=mat)

        layer = N.Layer([make() for _ in range(n_nodes)])
        flow += [sb, layer]
        w, h = sb.out_channels_xy
        c = k + q
        if w * h == 1:
            break
    return flow


def remainder_net(seed=0):
    """32x32 input, four layers; layers 2 and 3 have 17..20 and 33..36 outputs per affine with more than four
    nodes, i.e. they take the 4x4-MFMA remainder-tile instantiations of k_stage (2x2 and 3x3 tiles)."""
    rng = np.random.default_rng(seed)
    funcs = [N.identity, N.unsigned_08expo]

    def layer(n, d_in, p, s):
        out = []
        for _ in range(n):
            ex = N.GeneralExpansionNode(funcs, p)
            out.append(N.FlowNode([rand_pca(rng, d_in, p, N.WhiteningNode), ex, rand_sfa(rng, ex.output_dim, s)]))
        return N.Layer(out)

    sb0 = N.Rectangular2dSwitchboard((32, 32), (4, 4), (4, 4), 1)      # 64 nodes
    sb1 = N.Rectangular2dSwitchboard((8, 8), (2, 1), (2, 1), 9)        # 32 nodes of 18 inputs
    sb2 = N.Rectangular2dSwitchboard((4, 8), (1, 2), (1, 2), 12)       # 16 nodes of 24 inputs
    sb3 = N.Rectangular2dSwitchboard((4, 4), (2, 1), (2, 1), 18)       # 8 nodes of 36 inputs
    return [sb0, layer(64, 16, 10, 9), sb1, layer(32, 18, 14, 12), sb2, layer(16, 24, 19, 18), sb3, layer(8, 36, 35, 34)]


def subtree_fuzz_net(seed):
    """Deep hierarchies WITHOUT overlap (every node read by exactly one node of the next layer), random merge shapes (2x1, 1x2, 2x2,
    3x1), node widths and expansions: the upper layers fall into independent sub-trees, which the fused plan may run as one launch
    per run of layers for short batches (k_subtree, plan_subtree).  Grids are chosen so that several layers have >= 4 nodes."""
    rng = np.random.default_rng(7000 + seed)
    w, h = [(16, 8), (8, 8), (12, 6), (16, 16), (8, 4), (6, 6)][int(rng.integers(0, 6))]
    pool = [N.identity, N.unsigned_08expo, N.signed_08expo, N.unsigned_expo(float(rng.uniform(0.5, 1.5)))]
    fx0, fy0 = (int(v) for v in rng.choice([2, 3, 4], 2))
    sb = N.Rectangular2dSwitchboard((w * fx0, h * fy0), (fx0, fy0), (fx0, fy0), 1)
    flow, c = [], 1
    first = True
    while True:
        if first:
            first = False
        else:
            opts = [(fx, fy) for fx, fy in ((2, 1), (1, 2), (2, 2), (3, 1), (1, 3)) if w % fx == 0 and h % fy == 0]
            if not opts:
                break
            fx, fy = opts[int(rng.integers(0, len(opts)))]
            sb = N.Rectangular2dSwitchboard((w, h), (fx, fy), (fx, fy), c)
        n_nodes, d_in = sb.output_channels, sb.out_channel_dim
        p0 = int(rng.integers(3, min(d_in, 48) + 1))
        s0 = int(rng.integers(4, 61))
        linear = rng.random() < 0.15
        nf = int(rng.integers(1, 3))
        funcs = [pool[int(i)] for i in rng.choice(len(pool), nf, replace=False)]

        def make():
            if linear:
                return N.FlowNode([rand_pca(rng, d_in, p0), rand_sfa(rng, p0, s0)])
            ex = N.GeneralExpansionNode(funcs, p0)
            return N.FlowNode([rand_pca(rng, d_in, p0), ex, rand_sfa(rng, ex.output_dim, s0)])

        layer = N.Layer([make() for _ in range(n_nodes)])
        flow += [sb, layer]
        w, h = sb.out_channels_xy
        c = layer.output_dim // n_nodes
        if w * h == 1:
            break
    return flow
