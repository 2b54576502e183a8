"""PCIe-inclusive rate of the drop-in call: host ndarray in -> Flow.execute -> host ndarray out
(hg_flow_execute: H2D, all kernels, D2H, synchronous).  DESIGN.md quotes this; it is never bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
for dt in (np.float64, np.float32, np.uint8):
    for n in (728, 4096):
        x = synth.make_subimages(n, 128, dtype=dt)
        f = Flow.from_blob(blob, output_dtype=np.float64)
        f.execute(x[:64])
        f.execute(x)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); y = f.execute(x, n_cols=20); best = min(best, time.perf_counter() - t0)
        print("host path  %-8s N=%5d  %8.2f ms  %9.0f sub-images/s  (%.1f GB/s of input over PCIe)" % (
            np.dtype(dt).name, n, best * 1e3, n / best, x.nbytes / best / 1e9))
        f.close()
