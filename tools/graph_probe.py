"""Does a captured graph of one call replay faster than its launches?  (round 5 what-if)  python tools/graph_probe.py [rows ...]
Round 5, one box: N = 1 / 18 / 130: 73.9 / 77.0 / 102.8 us per replay against 69.4 / 72.2 / 97.5 us per call of plain launches (same
results) — a graph does not shorten a short call.  From a few hundred rows on a replay is NOT the same computation: the front kernel's tile
queues count up from a base that is a kernel ARGUMENT of each launch (never reset, hg_fused.hip: work_counters), and a replay repeats the
captured base — tiles are skipped, the replay is "faster" and wrong.  The library does not use graphs."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
flow.reserve(4096)
xs = torch.from_numpy(synth.make_subimages(4096, 128, dtype=np.uint8)).to(dev)
s = torch.cuda.Stream(dev)
for n in [int(a) for a in sys.argv[1:]] or (1, 18, 130, 348, 1738, 4096):
    x = xs[:n]
    y = torch.empty((n, 20), dtype=torch.float32, device=dev)
    yg = torch.empty((n, 20), dtype=torch.float32, device=dev)
    def call(out, st):
        flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], out.data_ptr(), np.float32, 20, 20, stream=st)
    with torch.cuda.stream(s):
        for _ in range(50):
            call(y, s.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            call(y, s.cuda_stream)
        torch.cuda.synchronize()
        plain = (time.perf_counter() - t0) / 500 * 1e6
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            call(yg, s.cuda_stream)
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 500 * 1e6
        same = bool(torch.equal(y, yg))
        print("N=%5d  launches %.1f us per call   graph replay %.1f us   same results: %s" % (n, plain, graph, same), flush=True)
    except Exception as e:
        print("N=%5d  launches %.1f us per call   capture failed: %s" % (n, plain, str(e)[:200]), flush=True)
