"""Latency of one device-resident call for small batches (the reference calls execute once per pyramid
level and cascade stage: N = 1..728 on a real frame).  Prints wall us per call (back-to-back async calls,
one sync at the end) and the sum of the per-stage kernel times from the library's profiling events."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
for n in (1, 16, 64, 256, 728, 1738):
    x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.float32)).to(dev)
    y = torch.empty((n, 20), dtype=torch.float32, device=dev)
    flow.reserve(n)
    def call(profile=False):
        flow.execute_device(x.data_ptr(), np.dtype(np.float32), n, x.shape[1], y.data_ptr(), np.float32, 20, 20,
                            stream=stream.cuda_stream, profile=profile)
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
        torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / reps * 1e6
    call(True); torch.cuda.synchronize()
    ks = sum(ms / max(c, 1) for _, ms, c in flow.stage_times()) * 1e3
    print("N=%5d: %.1f us per call back-to-back, %.1f us call+sync, kernels %.1f us" % (n, wall, lat, ks), flush=True)

# the same call replayed from a captured graph (what a caller with stable buffers can do today)
for n in (1, 64, 728, 1738):
    x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.float32)).to(dev)
    y = torch.empty((n, 20), dtype=torch.float32, device=dev)
    flow.reserve(n)
    s = torch.cuda.Stream(dev)
    def call_on(st):
        flow.execute_device(x.data_ptr(), np.dtype(np.float32), n, x.shape[1], y.data_ptr(), np.float32, 20, 20, stream=st.cuda_stream)
    with torch.cuda.stream(s):
        for _ in range(3):
            call_on(s)
    torch.cuda.synchronize()
    y_ref = y.clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            call_on(s)
    except Exception as e:
        print("capture failed:", e)
        break
    y.zero_()
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(y, y_ref)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / reps * 1e6
    print("N=%5d graph replay: %.1f us back-to-back, %.1f us replay+sync, identical output: %s" % (n, wall, lat, ok), flush=True)
