"""Latency of one device-resident call at small batch sizes with / without the persistent top-of-hierarchy launch,
and bit-identity of the two (HIGSFA_CHAIN_MAX_TILES is read at plan time)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
flows = {}
for name, mt in (("chain", "64"), ("layers", "0")):
    os.environ["HIGSFA_CHAIN_MAX_TILES"] = mt
    flows[name] = Flow.from_blob(blob, device=0, output_dtype=np.float32)
    flows[name].info()
print(flows["chain"].describe()[-900:])
for n in (1, 16, 33, 98, 340, 728, 1024, 1738):
    x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.uint8)).to(dev)
    ys = {}
    line = "N=%5d:" % n
    for name, flow in flows.items():
        y = torch.empty((n, 60), dtype=torch.float32, device=dev)
        flow.reserve(max(n, 2048))
        def call(profile=False):
            flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream, profile=profile)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.15:      # leave the idle power state (bench.py: settle)
            for _ in range(20):
                call()
            torch.cuda.synchronize()
        reps = 1000
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e6
        call(True); torch.cuda.synchronize()
        st = flow.stage_times()
        ks = sum(ms / max(c, 1) for _, ms, c in st) * 1e3
        top = sum(ms / max(c, 1) for _, ms, c in st[6:11]) * 1e3
        ys[name] = y.cpu().numpy().copy()
        line += "  %s %.1f us/call (kernels %.1f, layers 6-10 %.1f)" % (name, wall, ks, top)
    line += "  identical: %s" % np.array_equal(ys["chain"], ys["layers"])
    print(line, flush=True)
