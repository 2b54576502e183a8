"""Throughput at N = 4096 with several independent batches in flight (one flow handle and stream each), against one."""
import os, sys, time, threading
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.float32)).to(dev)
K = 1000
for n_par in (1, 2, 3):
    flows = [Flow.from_blob(blob, device=0, output_dtype=np.float32) for _ in range(n_par)]
    ys = [torch.empty((n, 20), dtype=torch.float32, device=dev) for _ in range(n_par)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_par)]
    for f in flows:
        f.reserve(n)

    def run(k):
        for s in range(k):
            i = s % n_par
            flows[i].execute_device(x.data_ptr(), np.dtype(np.float32), n, 16384, ys[i].data_ptr(), np.float32, 20, 20, stream=streams[i].cuda_stream)
        torch.cuda.synchronize()
    run(150)
    t0 = time.perf_counter()
    run(K)
    dt = time.perf_counter() - t0
    print("%d in flight: %.4f ms per step, %.0f sub-images/s, same output: %s" % (n_par, dt / K * 1e3, n * K / dt, all(torch.equal(ys[0], y) for y in ys)), flush=True)
    for f in flows:
        f.close()
