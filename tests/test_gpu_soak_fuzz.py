"""GPU: the two robustness tools of rounds 2-3 as tests (they ran by hand only).

* soak (tools/soak.py, 10 s here): device-resident calls with random batch sizes (1 .. 4096), input types and row offsets
  through ONE flow handle; every result bit-equal to the rows of one N = 4096 reference (on U11L-128 a row's result does not
  depend on the batch it travels in), and a failed internal hand-off of the front kernel's tile queue would
  surface as an exception of the next call.
* wide fuzz (tools/fuzz_wide.py, 50 hierarchies here): random ordinary / product / iGSFA hierarchies, full output and a
  random n_cols, against the float64 oracle (tolerance of BASELINE.json: 1e-4 of max|ref|)."""
import time

import numpy as np
import pytest

from oracle import mdp_restate as oracle
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

pytestmark = pytest.mark.gpu


def test_soak_ten_seconds(native_lib):
    import torch
    from pyfaceanalysis_amd import synth
    blob, nodes = synth.cached_preset_blob("U11L-128")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
    flow.reserve(4096)
    x8 = torch.from_numpy(synth.make_subimages(4096, 128, dtype=np.uint8)).to(dev)
    xs = {np.uint8: x8, np.float32: x8.float(), np.float64: x8.double()}
    ref = torch.empty((4096, 60), dtype=torch.float32, device=dev)
    flow.execute_device(x8.data_ptr(), np.dtype(np.uint8), 4096, 16384, ref.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    y = torch.empty((4096, 60), dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    calls = rows = bad = 0
    while time.perf_counter() - t0 < 10.0:
        for _ in range(50):
            n = int(rng.choice([1, 7, 16, 17, 100, 128, 129, 340, 728, 1000, 1738, 2048, 4095, 4096]))
            dt = [np.uint8, np.float32, np.float64][int(rng.integers(0, 3))]
            off = int(rng.integers(0, 4096 - n + 1))
            x = xs[dt][off:off + n]
            flow.execute_device(x.data_ptr(), np.dtype(dt), n, 16384, y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
            bad += not torch.equal(y[:n], ref[off:off + n])
            calls += 1
            rows += n
        torch.cuda.synchronize()
    print("soak: %d calls, %d rows in %.1f s, mismatching calls: %d" % (calls, rows, time.perf_counter() - t0, bad))
    flow.close()
    assert bad == 0 and calls > 1000


def test_host_path_soak_five_seconds(native_lib):
    """The ndarray-in / ndarray-out call for 5 s with random batch sizes, types and row offsets: packers storing straight into
    the pass buffers in device memory (write-combined stores, fence, flag, launch from another thread), six buffers rotating,
    features stored into pinned memory by the last kernel and polled — a row left in a write-combining buffer, a kernel reading
    a stale line of a reused buffer or a feature row read before it arrived would show as a mismatch against the rows of one
    device-resident reference."""
    import torch
    from pyfaceanalysis_amd import synth
    blob, nodes = synth.cached_preset_blob("U11L-128")
    flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
    x8 = synth.make_subimages(4096, 128, dtype=np.uint8)
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x8).to(dev)
    ref_d = torch.empty((4096, 20), dtype=torch.float32, device=dev)
    flow.execute_device(xd.data_ptr(), np.dtype(np.uint8), 4096, 16384, ref_d.data_ptr(), np.float32, 20, 20, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    ref = ref_d.cpu().numpy()
    xs = {np.uint8: x8, np.float32: x8.astype(np.float32), np.float64: x8.astype(np.float64)}
    rng = np.random.default_rng(3)
    t0 = time.perf_counter()
    calls = bad = 0
    while time.perf_counter() - t0 < 5.0:
        n = int(rng.choice([1, 5, 16, 17, 100, 129, 340, 728, 1000, 1738, 2049, 4096]))
        dt = [np.uint8, np.float32, np.float64][int(rng.integers(0, 3))]
        off = int(rng.integers(0, 4096 - n + 1))
        y = flow.execute(xs[dt][off:off + n], n_cols=20)
        bad += not np.array_equal(y, ref[off:off + n])
        calls += 1
    print("host-path soak: %d calls in %.1f s, mismatching calls: %d" % (calls, time.perf_counter() - t0, bad))
    flow.close()
    assert bad == 0 and calls > 500


def test_wide_fuzz_fifty_hierarchies(native_lib, monkeypatch):
    bad, tails, subs, n = [], 0, 0, 0
    for kind, maker, seeds in (("net", helpers.fuzz_net, range(100, 130)), ("prod", helpers.fuzz_product_net, range(100, 112)),
                               ("igsfa", helpers.fuzz_igsfa_net, range(100, 108))):
        for seed in seeds:
            nodes = maker(seed)
            rng = np.random.default_rng(seed)
            x = rng.normal(size=(int(rng.integers(1, 70)), nodes[0].input_dim)) * 1.5
            f = Flow(nodes)
            tails += "no unpack pass" in f.describe()
            y = f.execute(x)
            if "sub-trees in ONE launch" in f.describe():      # layers run as sub-trees for these few rows: same bits as per-layer launches
                subs += 1
                monkeypatch.setenv("HIGSFA_SUBTREE", "0")
                g = Flow(nodes)
                same = "sub-trees in ONE launch" not in g.describe() and np.array_equal(g.execute(x), y)
                monkeypatch.delenv("HIGSFA_SUBTREE")
                g.close()
                if not same:
                    bad.append((kind, seed, "sub-tree launch differs", f.info().plan_kind))
            k = int(rng.integers(1, nodes[-1].output_dim + 1))
            yk = f.execute(x, n_cols=k)
            ref = oracle.execute_flow(nodes, x)
            err = np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)
            if not (err <= 1e-4 and np.array_equal(yk, y[:, :k])):
                bad.append((kind, seed, float(err), f.info().plan_kind))
            n += 1
            f.close()
    print("wide fuzz: %d flows, %d with the top-of-hierarchy launch, %d with sub-tree launches, mismatches: %r" % (n, tails, subs, bad))
    assert n == 50 and not bad
