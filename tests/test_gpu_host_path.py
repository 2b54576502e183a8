"""GPU: the ndarray-in / ndarray-out call (hg_flow_execute, FaceDetectUpdated.py:699) through the host pipeline
(hg_hostpipe.hpp: packers narrowing exactly / copying, straight into device memory on large-BAR devices or through the
pinned ring and the copy queues; passes sized by the planner), and the same call over several replicas
(hg_flow_execute_sharded)."""
import numpy as np
import pytest

from oracle import mdp_restate as oracle
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_err(y, ref):
    return float(np.abs(np.asarray(y, dtype=np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("direct", ["1", "0"])
def test_pipelined_chunks_narrowing_and_fallthrough(native_lib, nets, monkeypatch, direct):
    """Several passes per call (70 000 rows of this 256-pixel net: the pass buffers hold 65 536 uint8 rows, 8192 float64
    rows), so pass buffers are reused.  Integer pixels cross PCIe as uint8; from the first row that holds anything else (a
    fraction, a negative, 256, NaN) the rest of the call goes in its own type; either way every row equals what the plain
    uint8 / float64 call gives.  Both transports: stores into device memory (large BAR) and pinned ring + copy queues."""
    nodes = nets("T5L-16")
    n = 70000
    xi = synth.make_subimages(n, 16, dtype=np.uint8)
    monkeypatch.setenv("HIGSFA_HOST_DIRECT", direct)          # read once, when a flow is loaded
    flow = Flow(nodes)
    y8 = flow.execute(xi)
    idx = np.arange(0, n, 997)
    assert rel_err(y8[idx], oracle.execute_flow(nodes, xi[idx])) <= TOL
    for dt in (np.float64, np.float32):
        assert np.array_equal(flow.execute(xi.astype(dt)), y8)
    # a non-image value in the middle and in the last row: the call falls through to the wide type there, no other row changes
    xf = xi.astype(np.float64)
    xf[20000, 5] = 17.5
    xf[n - 1, 255] = -3.0
    xf[20001, 7] = 256.0
    yf = flow.execute(xf)
    touched = np.zeros(n, dtype=bool)
    touched[[20000, 20001, n - 1]] = True
    assert np.array_equal(yf[~touched], y8[~touched])
    assert rel_err(yf[touched], oracle.execute_flow(nodes, xf[touched])) <= TOL
    # narrowing switched off: same bits
    monkeypatch.setenv("HIGSFA_NO_NARROW", "1")          # read once, when a flow is loaded
    wide_flow = Flow(nodes)
    monkeypatch.delenv("HIGSFA_NO_NARROW")
    assert np.array_equal(wide_flow.execute(xi.astype(np.float64)), y8)
    wide_flow.close()
    # strided rows (ldx > input_dim) and the first-k-columns form through the same staging
    wide = np.zeros((n, 260), dtype=np.float32)
    wide[:, 2:258] = xi
    assert np.array_equal(flow.execute(wide[:, 2:258], n_cols=3), y8[:, :3])
    flow.close()


@pytest.mark.parametrize("direct", ["1", "0"])
def test_u11l_host_path_all_dtypes(native_lib, nets, monkeypatch, direct):
    """The reference's own call shape: float64 ndarray of 128x128 pixel values in, float64 features out, at the
    largest batch a real frame produces (N = 728, SURVEY.md §6: two passes), at sizes around the planner's pass
    boundaries and at a size that needs more passes than there are pass buffers (6): every call bit-equal to the rows of
    the device-resident call on the same pixels, whatever the type handed over."""
    nodes = nets("U11L-128")
    monkeypatch.setenv("HIGSFA_HOST_DIRECT", direct)
    flow = Flow(nodes)
    x8 = synth.make_subimages(5000, 128, dtype=np.uint8)
    y = flow.execute(x8[:728].astype(np.float64))
    assert y.dtype == np.float64 and y.shape == (728, 60)
    assert np.array_equal(y, flow.execute(x8[:728])) and np.array_equal(y, flow.execute(x8[:728].astype(np.float32)))
    idx = np.arange(0, 728, 29)
    assert rel_err(y[idx], oracle.execute_flow(nodes, x8[idx])) <= TOL
    import torch
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x8).to(dev)
    yd = torch.empty((5000, 60), dtype=torch.float64, device=dev)
    flow.execute_device(xd.data_ptr(), np.uint8, 5000, 16384, yd.data_ptr(), np.float64, 60, 60, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    ref = yd.cpu().numpy()
    for n, dt in ((1, np.float64), (7, np.uint8), (17, np.float32), (300, np.float64), (1023, np.uint8), (1025, np.float64), (5000, np.float64), (5000, np.uint8)):
        assert np.array_equal(flow.execute(x8[:n].astype(dt)), ref[:n]), (n, dt)
    xf = x8[:3000].astype(np.float32)
    xf[1500, 77] = 0.25                          # falls through to float32 from the pass that holds row 1500
    yf = flow.execute(xf)
    keep = np.arange(3000) != 1500
    assert np.array_equal(yf[keep], ref[:3000][keep]) and not np.array_equal(yf[1500], ref[1500])
    flow.close()


def test_direct_stores_only_into_confirmed_host_mapped_buffers(native_lib, nets, monkeypatch):
    """the packers store into a pass's input buffer only after hsa_amd_pointer_info has confirmed THAT
    allocation host-mapped at its device address; any other answer — forced here with HIGSFA_HOST_PROBE_DENY, read per probe —
    selects the pinned ring, with the same bits.  On a box whose device has no large BAR both calls take the ring."""
    nodes = nets("T5L-16")
    x = synth.make_subimages(9000, 16, dtype=np.float64)
    flow = Flow(nodes)
    assert flow.host_transport() == -1
    y = flow.execute(x)
    t_default = flow.host_transport()
    assert t_default in (0, 1)
    monkeypatch.setenv("HIGSFA_HOST_PROBE_DENY", "1")
    denied = Flow(nodes)                      # fresh handle: fresh buffers, probed under the denial
    yd = denied.execute(x)
    assert denied.host_transport() == 0
    assert np.array_equal(yd, y)
    monkeypatch.delenv("HIGSFA_HOST_PROBE_DENY")
    assert np.array_equal(denied.execute(x[:4000]), y[:4000]) and denied.host_transport() == 0      # the answer is kept per allocation
    # both transports on a call with several passes and a narrowing fall-through in the middle
    xf = x.copy()
    xf[5000, 3] = 0.5
    assert np.array_equal(denied.execute(xf), flow.execute(xf)) and flow.host_transport() == t_default
    flow.close()
    denied.close()


def test_sharded_c_entry_on_replicas(native_lib, nets):
    """hg_flow_execute_sharded with the one visible GPU listed two and three times: every listed entry is a replica
    with its own weights, streams and staging, the row blocks run concurrently from separate host threads and land
    in the caller's matrix at their own rows.  Same bits as the single-device call, ragged and tiny batches too."""
    nodes = nets("T5L-16")
    flow = Flow(nodes)
    x = synth.make_subimages(1001, 16, dtype=np.float64)
    y1 = flow.execute(x)
    for devs in ([0], [0, 0], [0, 0, 0]):
        for n in (1001, 1, 2, 17, 0):
            y = flow.execute(x[:n], devices=devs)
            assert y.shape == (n, 10) and np.array_equal(y, y1[:n]), (devs, n)
    assert np.array_equal(flow.execute(x, devices=[0, 0], n_cols=4), y1[:, :4])
    with pytest.raises(RuntimeError, match="out of range"):
        flow.execute(x, devices=[0, 99])
    flow.close()
    # the 11-layer net, two replicas, uint8 input
    nodes = nets("U11L-128")
    flow = Flow(nodes, output_dtype=np.float32)
    x8 = synth.make_subimages(300, 128, dtype=np.uint8)
    assert np.array_equal(flow.execute(x8, devices=[0, 0], n_cols=20), flow.execute(x8, n_cols=20))
    flow.close()


def test_batch_size_dependence_is_bounded(native_lib, nets, monkeypatch):
    """What DESIGN.md §3.1 says about N.  The reference's caller re-feeds survivors at other batch sizes
    (FaceDetectUpdated.py:755), so a row's result should not depend on the batch it arrives in.  With the one-tile-per-pass
    front kernel (the default for U11L-128) every batch size takes the same kernels in the same summation order: results
    are bit-identical across N.  The generic front kernel (two tiles per pass; any expansion) hands single-tile batches to
    the unfused first-layer kernels, whose summation order differs in the last bits: that difference is pinned at
    <= 2e-6 of max|y|, far inside the 1e-4 budget."""
    nodes = nets("U11L-128")
    x = synth.make_subimages(4096, 128, dtype=np.uint8)
    flow = Flow(nodes, output_dtype=np.float32)
    big = flow.execute(x, n_cols=20)
    for n in (16, 7, 1, 32, 100, 129, 728, 1738):
        assert np.array_equal(flow.execute(x[:n], n_cols=20), big[:n]), n
    flow.close()
    monkeypatch.setenv("HIGSFA_NO_FSPEC", "1")
    generic = Flow(nodes, output_dtype=np.float32)
    monkeypatch.delenv("HIGSFA_NO_FSPEC")
    gbig = generic.execute(x, n_cols=20)
    assert np.array_equal(gbig, big)
    for n in (16, 7, 32, 728):
        small = generic.execute(x[:n], n_cols=20)
        d = float(np.abs(small.astype(np.float64) - gbig[:n]).max() / np.abs(gbig).max())
        assert d <= 2e-6, (n, d)
    assert np.array_equal(generic.execute(x[:728], n_cols=20), generic.execute(x[:728], n_cols=20))
    generic.close()


def test_top_of_hierarchy_launch(native_lib, nets, monkeypatch):
    """The last layers of U11L-128 (4, 2, 1 nodes) run as ONE launch that keeps the activations in LDS and writes the
    caller's rows itself (k_tail, hg_fused_tail.hip; HIGSFA_TAIL = how many layers it may fuse, 0 = per-layer launches +
    k_unpack).  Same products in the same order as the per-layer kernels: bit-identical features for every fusion depth,
    batch size, column count, output dtype and row stride."""
    nodes = nets("U11L-128")
    x = synth.make_subimages(1100, 128, dtype=np.uint8)
    flows = {}
    for depth in ("0", "1", "2", "3"):
        monkeypatch.setenv("HIGSFA_TAIL", depth)
        flows[depth] = Flow(nodes, output_dtype=np.float32)
        desc = flows[depth].describe()
        assert ("no unpack pass" in desc) == (depth != "0")
        assert ("ONE launch, activations in LDS" in desc) == (depth in ("2", "3"))
    monkeypatch.delenv("HIGSFA_TAIL")
    default = Flow(nodes, output_dtype=np.float32)
    assert "ONE launch, activations in LDS" in default.describe()
    base = flows["0"].execute(x)
    assert base.shape == (1100, 60)
    ref = oracle.execute_flow(nodes, x[:33])
    assert rel_err(base[:33], ref) <= TOL
    for name, f in list(flows.items()) + [("default", default)]:
        for n in (1, 16, 17, 128, 100, 1, 300, 33, 1100):
            assert np.array_equal(f.execute(x[:n]), base[:n]), (name, n)
        for cols in (1, 9, 16, 17, 20, 33, 59):                 # only the output tiles that hold a requested column are computed
            assert np.array_equal(f.execute(x[:50], n_cols=cols), base[:50, :cols]), (name, cols)
    # float64 out (the MDP default) and a strided y through the device entry
    f64 = Flow(nodes)
    assert np.array_equal(f64.execute(x[:77], n_cols=20), base[:77, :20].astype(np.float64))
    import torch
    xd = torch.from_numpy(x[:200]).cuda()
    yd = torch.full((200, 32), -7.0, dtype=torch.float32, device="cuda")
    default.execute_device(xd.data_ptr(), np.uint8, 200, xd.shape[1], yd.data_ptr(), np.float32, 20, 32)
    torch.cuda.synchronize()
    yh = yd.cpu().numpy()
    assert np.array_equal(yh[:, :20], base[:200, :20]) and np.all(yh[:, 20:] == -7.0)      # columns beyond y_cols untouched
    for f in list(flows.values()) + [default, f64]:
        f.close()


@pytest.mark.parametrize("maker", ["linear", "overlap", "u11l64", "linear96", "narrow_top_long_batch"])
def test_top_of_hierarchy_launch_other_nets(native_lib, nets, monkeypatch, maker):
    """k_tail on other layer shapes: a linear top (no expansion), uneven node widths, the 64x64 preset, the linear 96x96 age-net
    shape — against the per-layer kernels (bit for bit) and the oracle.  narrow_top_long_batch: the TWO-tiles-per-workgroup
    instantiation (k_tail<2, .>: tops of at most 8 waves from 512 tiles on; ADVICE r3: no test reached it) on T5L-16 (4-2-1 nodes
    of one m-tile: 4 waves) with 8200 rows = 513 tiles, an odd count, so the last workgroup holds one real tile and one beyond
    the batch."""
    nodes = {"linear": lambda: helpers.linear_net(3), "overlap": lambda: helpers.overlapping_net(5), "u11l64": lambda: nets("U11L-64"),
             "linear96": lambda: helpers.linear_u11l_96(1), "narrow_top_long_batch": lambda: nets("T5L-16")}[maker]()
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (8200 if maker == "narrow_top_long_batch" else 150, nodes[0].input_dim)).astype(np.float32)
    monkeypatch.setenv("HIGSFA_TAIL", "0")
    per_layer = Flow(nodes, output_dtype=np.float32)
    monkeypatch.delenv("HIGSFA_TAIL")
    fused = Flow(nodes, output_dtype=np.float32)
    a, b = per_layer.execute(x), fused.execute(x)
    assert np.array_equal(a, b)
    k = max(1, nodes[-1].output_dim // 3)
    assert np.array_equal(fused.execute(x[:19], n_cols=k), a[:19, :k])
    idx = np.arange(0, len(x), max(1, len(x) // 150))
    assert rel_err(b[idx], oracle.execute_flow(nodes, x[idx])) <= TOL
    print(maker, [ln for ln in fused.describe().splitlines() if "launch" in ln or "unpack" in ln][-2:])
    per_layer.close()
    fused.close()


@pytest.mark.parametrize("preset", ["U11L-128", "U11L-64"])
def test_subtree_launch_for_short_batches(native_lib, nets, monkeypatch, preset):
    """Short batches (up to 512 rows; a frame's later cascade stages hold 18 .. 348 windows) run layers 6-8 of the 11-layer nets
    (16, 8, 4 nodes) as four independent sub-trees in ONE launch and layers 3-5 (128, 64, 32 nodes) as 32 (k_subtree,
    hg_fused_tail.hip; HIGSFA_SUBTREE = largest batch in 16-row tiles, 0 = never).  Same products in the same order as the per-layer
    kernels: the same bits with the launches off, at their default and forced on for long batches (where the run under the top
    takes the top launch's first layer away from it), for ragged batches and for both output types; and the sub-trees the planner
    found are the roots' own (the nodes a root reads are not consecutive in every layer).  Between 130 and 512 rows the default plan
    uses its alternative set of runs (layers 5-7 as eight sub-trees): the batch sizes below cover both sets."""
    nodes = nets(preset)
    side = 128 if preset == "U11L-128" else 64
    x = synth.make_subimages(1300, side, dtype=np.uint8)
    monkeypatch.setenv("HIGSFA_SUBTREE", "0")
    off = Flow(nodes, output_dtype=np.float32)
    assert "sub-trees in ONE launch" not in off.describe()      # (the plan is made, and the environment read, at first use)
    monkeypatch.setenv("HIGSFA_SUBTREE", "100000")
    monkeypatch.setenv("HIGSFA_SUBTREE_WGS", "100000000")
    forced = Flow(nodes, output_dtype=np.float32)
    d_forced = forced.describe()
    monkeypatch.delenv("HIGSFA_SUBTREE")
    monkeypatch.delenv("HIGSFA_SUBTREE_WGS")
    default = Flow(nodes, output_dtype=np.float32)
    for d in (d_forced, default.describe()):
        assert "2 layer(s) as 4 sub-trees in ONE launch" in d and d.count("[in the sub-tree launch for short batches]") >= 3, d
    base = off.execute(x)
    assert rel_err(base[:40], oracle.execute_flow(nodes, x[:40])) <= TOL
    for name, f in (("default", default), ("forced", forced)):
        for n in (1, 16, 17, 18, 44, 130, 348, 512, 513, 1, 1300, 700):
            assert np.array_equal(f.execute(x[:n]), base[:n]), (name, n)
        assert np.array_equal(f.execute(x[:130], n_cols=20), base[:130, :20]), name
    f64 = Flow(nodes)
    assert np.array_equal(f64.execute(x[:348], n_cols=20), base[:348, :20].astype(np.float64))
    # benchmark= timings: one entry per stage as ever; a run's time is carried by the first of its three layers
    class Bench(object):
        enabled = True

        def __init__(self):
            self.tasks = []

        def add_task_ellapsed(self, label, secs, reference=None):
            self.tasks.append((label, secs))

    b = Bench()
    assert np.array_equal(default.execute(x[:130], benchmark=b), base[:130])
    assert len(b.tasks) == default.info().n_stages
    # (at 130 rows one three-layer run is in use — of the alternative set, eight sub-trees, where the net has one: its time is carried by
    # the first of its layers, the other two read the few microseconds between two event records)
    assert all(t[1] >= 0 for t in b.tasks) and sum(t[1] for t in b.tasks) > 0
    assert "layers from here as 8 sub-trees]" in default.describe()
    for f in (off, forced, default, f64):
        f.close()


@pytest.mark.parametrize("seed", range(16))
def test_subtree_launch_on_random_tree_hierarchies(native_lib, monkeypatch, seed):
    """Random hierarchies without overlap (helpers.subtree_fuzz_net: merges of 2, 3 and 4 children, 4 ... 32 roots, node widths of one
    to four tiles, one or two expansion functions or none): where the planner finds runs of sub-trees, short batches through them
    give the bits of per-layer launches (HIGSFA_SUBTREE=0) and agree with the oracle."""
    nodes = helpers.subtree_fuzz_net(seed)
    rng = np.random.default_rng(seed)
    x = (rng.normal(size=(150, nodes[0].input_dim)) * 1.5).astype(np.float32)
    monkeypatch.setenv("HIGSFA_SUBTREE", "0")
    off = Flow(nodes, output_dtype=np.float32)
    assert "sub-trees in ONE launch" not in off.describe()
    monkeypatch.delenv("HIGSFA_SUBTREE")
    on = Flow(nodes, output_dtype=np.float32)
    planned = "sub-trees in ONE launch" in on.describe()
    base = off.execute(x)
    assert rel_err(base[:24], oracle.execute_flow(nodes, x[:24])) <= TOL
    for n in (1, 16, 20, 75, 150, 130):
        assert np.array_equal(on.execute(x[:n]), base[:n]), (seed, n, planned)
    k = max(1, nodes[-1].output_dim // 2)
    assert np.array_equal(on.execute(x[:33], n_cols=k), base[:33, :k])
    print("seed %d: %s" % (seed, [ln.split("[batches of up to")[1][:64] for ln in on.describe().splitlines() if "sub-trees in ONE launch" in ln]))
    off.close()
    on.close()


def test_subtree_launch_is_planned_only_where_the_layers_split(native_lib, nets):
    """Overlapping receptive fields (a node of the layer below feeds two nodes above) leave no independent sub-trees: no plan."""
    for nodes in (helpers.overlapping_net(5), helpers.linear_net(3), nets("T5L-16")):
        f = Flow(nodes, output_dtype=np.float32)
        d = f.describe()
        x = np.random.default_rng(5).integers(0, 256, (40, nodes[0].input_dim)).astype(np.float32)
        assert rel_err(f.execute(x), oracle.execute_flow(nodes, x)) <= TOL
        if "sub-trees in ONE launch" in d:      # (a net whose upper layers do split: at least four roots)
            assert int(d.split(" sub-trees in ONE launch")[0].split(" as ")[-1]) >= 4
        f.close()


def test_front_kernel_variants_agree(native_lib, nets, monkeypatch):
    """Layers 0+1 of U11L-128 run on one of four kernels: every wave on its own with direct loads (k_stage01d, the default
    for this layout; with one tile queue per chunk shared through an LDS ring or, HIGSFA_NO_WGQ, one per layer-1 node), the LDS-staged form with the expansion known at compile time (HIGSFA_NO_DIRECT), and the generic
    LDS-staged form (HIGSFA_NO_FSPEC).  All three hand tiles out dynamically (work counters that are never reset) and
    must give the same bits for every input type, for ragged batches, and call after call."""
    nodes = nets("U11L-128")
    x8 = synth.make_subimages(1000, 128, dtype=np.uint8)
    direct = Flow(nodes, output_dtype=np.float32)
    monkeypatch.setenv("HIGSFA_NO_WGQ", "1")              # k_stage01d with one tile queue per layer-1 node
    direct_wg = Flow(nodes, output_dtype=np.float32)
    monkeypatch.delenv("HIGSFA_NO_WGQ")
    monkeypatch.setenv("HIGSFA_NO_DIRECT", "1")
    staged = Flow(nodes, output_dtype=np.float32)
    monkeypatch.setenv("HIGSFA_NO_FSPEC", "1")
    generic = Flow(nodes, output_dtype=np.float32)
    monkeypatch.delenv("HIGSFA_NO_DIRECT")
    monkeypatch.delenv("HIGSFA_NO_FSPEC")
    ref = None
    for n in (1000, 16, 17, 333, 1, 1000, 640):
        for dt in (np.uint8, np.float32, np.float64):
            x = x8[:n].astype(dt)
            a = direct.execute(x)
            assert np.array_equal(a, staged.execute(x)) and np.array_equal(a, generic.execute(x)) and np.array_equal(a, direct_wg.execute(x)), (n, dt)
            if n == 1000:
                ref = a if ref is None else ref
                assert np.array_equal(a, ref)
    assert rel_err(direct.execute(x8[:64]), oracle.execute_flow(nodes, x8[:64])) <= TOL
    for f in (direct, direct_wg, staged, generic):
        f.close()


def test_tile_queue_counters_wrap_around(native_lib, nets, monkeypatch):
    """The front kernels hand tiles out through counters that are never reset: the host only tracks their base, modulo 2^32
    (StageParams::work_ctr).  A process that serves frames for days crosses 2^32; here the counters start 300 below it, so
    the second call of 256 tiles per queue wraps — and every call, before, across and after, must give the usual bits."""
    nodes = nets("U11L-128")
    x = synth.make_subimages(4096, 128, dtype=np.uint8)
    usual = Flow(nodes, output_dtype=np.float32)
    ref = usual.execute(x)
    usual.close()
    for variant in ({}, {"HIGSFA_NO_WGQ": "1"}, {"HIGSFA_NO_DIRECT": "1"}, {"HIGSFA_NO_DIRECT": "1", "HIGSFA_NO_FSPEC": "1"}):
        monkeypatch.setenv("HIGSFA_WQ_START", "0xfffffed4")
        for k, v in variant.items():
            monkeypatch.setenv(k, v)
        flow = Flow(nodes, output_dtype=np.float32)
        for k in list(variant) + ["HIGSFA_WQ_START"]:
            monkeypatch.delenv(k)
        for n in (4096, 4096, 1000, 4096, 17, 4096):
            assert np.array_equal(flow.execute(x[:n]), ref[:n]), (variant, n)
        flow.close()


def test_device_resident_strided_rows(native_lib, nets):
    """hg_flow_execute_device with sub-images that are rows of a wider device matrix (ldx > input_dim): the front kernel
    addresses rows through the leading dimension (aligned ldx: fused layers-0+1 kernel with direct loads; ldx that breaks
    the 16-byte alignment of the rows: the unfused first-layer kernels).  Same features as the dense matrix, within the
    bound that holds across kernel choices."""
    import torch
    nodes = nets("U11L-128")
    n = 200
    x8 = synth.make_subimages(n, 128, dtype=np.uint8)
    flow = Flow(nodes, output_dtype=np.float32)
    dense = flow.execute(x8)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for np_dt, t_dt in ((np.float32, torch.float32), (np.uint8, torch.uint8), (np.float64, torch.float64)):
        for pad, exact in ((64, True), (3, False)):
            wide = torch.zeros((n, 16384 + pad), dtype=t_dt, device=dev)
            wide[:, :16384] = torch.from_numpy(x8.astype(np_dt)).to(dev)
            y = torch.empty((n, 60), dtype=torch.float32, device=dev)
            flow.execute_device(wide.data_ptr(), np.dtype(np_dt), n, wide.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            if exact:
                assert np.array_equal(got, dense), (np_dt, pad)
            else:
                assert float(np.abs(got.astype(np.float64) - dense).max() / np.abs(dense).max()) <= 2e-6, (np_dt, pad)
    flow.close()
