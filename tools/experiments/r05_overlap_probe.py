"""Opt-in overlapped launches (HIGSFA_OVERLAP=1: the second sub-tree launch of a short call on a side queue, held at a device-side counter):
bit-identity soak against plain launches over random short batches, then us per device-resident call both ways.
python tools/overlap_probe.py <soak seconds>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
import torch
blob, nodes = synth.cached_preset_blob("U11L-128")
x8 = synth.make_subimages(1024, 128, dtype=np.uint8)
ref_flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
ref = ref_flow.execute(x8, n_cols=20)
os.environ["HIGSFA_OVERLAP"] = "1"
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
flow.execute(x8[:18], n_cols=20)      # (the plan is made, and the environment read, at first use)
del os.environ["HIGSFA_OVERLAP"]
rng = np.random.default_rng(5)
bad = calls = 0
t0 = time.perf_counter()
sizes = {}
while time.perf_counter() - t0 < float(sys.argv[1]):
    n = int(rng.integers(1, 200)); off = int(rng.integers(0, 1024 - n + 1))
    y = flow.execute(x8[off:off + n], n_cols=20)
    if not np.array_equal(y, ref[off:off + n]):
        bad += 1; sizes[n] = sizes.get(n, 0) + 1
    calls += 1
print("overlap soak (host calls): %d calls, mismatching %d %r" % (calls, bad, dict(list(sizes.items())[:10])))
# device-resident back-to-back calls (the timing loop's pattern): correctness of the last call + time per call
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(dev)
xs = torch.from_numpy(x8).to(dev)
for f, name in ((ref_flow, "plain"), (flow, "overlap")):
    f.reserve(1024)
    line = []
    for n in (1, 18, 44, 100, 128):
        x = xs[:n]; y = torch.empty((n, 20), dtype=torch.float32, device=dev)
        def call():
            f.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 20, 20, stream=st.cuda_stream)
        for _ in range(200): call()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(500): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 500 * 1e6
        ok = np.array_equal(y.cpu().numpy(), ref[:n])
        line.append("%d: %.1f%s" % (n, dt, "" if ok else " WRONG"))
    print(name, "us per call ", "  ".join(line))
