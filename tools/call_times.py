"""Wall time of one device-resident call of U11L-128 over a range of batch sizes (steady state)."""
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
flow.reserve(4096)
xs = torch.from_numpy(synth.make_subimages(4096, 128, dtype=np.uint8)).to(dev)
line = []
for n in [int(a) for a in sys.argv[1:]] or (1, 16, 33, 98, 128, 200, 340, 500, 728, 1024, 1738, 2048, 3000, 4096):
    x = xs[:n]
    y = torch.empty((n, 60), dtype=torch.float32, device=dev)
    def call():
        flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.1:
        for _ in range(20):
            call()
        torch.cuda.synchronize()
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    line.append("%d: %.1f" % (n, (time.perf_counter() - t0) / reps * 1e6))
print("us per call  " + "  ".join(line), flush=True)
