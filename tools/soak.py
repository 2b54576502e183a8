"""Soak: random batch sizes and input types through one flow handle for a while; every result is compared with the
rows of a reference computed once at N = 4096 (results are bit-identical across batch sizes), and any failed internal
hand-off of the persistent kernels surfaces as an exception of the next call."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
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
while time.perf_counter() - t0 < seconds:
    for _ in range(50):
        n = int(rng.choice([1, 7, 16, 17, 100, 128, 129, 340, 728, 1000, 1738, 2048, 4095, 4096]))
        dt = [np.uint8, np.float32, np.float64][int(rng.integers(0, 3))]
        off = int(rng.integers(0, 4096 - n + 1))
        x = xs[dt][off:off + n]
        flow.execute_device(x.data_ptr(), np.dtype(dt), n, 16384, y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
        if not torch.equal(y[:n], ref[off:off + n]):
            bad += 1
        calls += 1
        rows += n
    torch.cuda.synchronize()
print("soak: %d calls, %d rows in %.1f s, mismatching calls: %d" % (calls, rows, time.perf_counter() - t0, bad), flush=True)
flow.close()
sys.exit(1 if bad else 0)
