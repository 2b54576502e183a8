"""Per-stage times of one device-resident call at the batch sizes a real frame produces."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
for n in (128, 340, 728, 1738):
    x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.uint8)).to(dev)
    y = torch.empty((n, 60), dtype=torch.float32, device=dev)
    flow.reserve(max(n, 2048))
    def call(profile=False):
        flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream, profile=profile)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:      # leave the idle power state (bench.py: settle)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 1000 * 1e6
    acc = None
    for _ in range(10):
        call(True); torch.cuda.synchronize()
        st = [ms / max(c, 1) * 1e3 for _, ms, c in flow.stage_times()]
        acc = st if acc is None else [min(a, b) for a, b in zip(acc, st)]
    print("N=%5d  %.1f us/call; stages (us, best of 10): %s  sum %.1f" % (n, wall, " ".join("%.1f" % v for v in acc), sum(acc)), flush=True)
