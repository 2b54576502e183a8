"""Flow.execute on HOST ndarrays (the call the reference makes, FaceDetectUpdated.py:699): ms per call over dtypes and batch
sizes, best of `reps` and median, beside the ceilings that bound it — the host reading the caller's array (float64 / float32
are narrowed to uint8 by the packing pool; tools/ubench/host_pack_bw.cpp measures that rate alone) and PCIe for the wire bytes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

HOST_READ_GBPS = float(os.environ.get("HOST_READ_GBPS", 288.0))      # 16 threads on the rows' memory node, narrowing float64
PCIE_GBPS = float(os.environ.get("PCIE_GBPS", 57.0))                 # pinned H2D, 64 MiB copies
blob, nodes = synth.cached_preset_blob("U11L-128")
f = Flow.from_blob(blob, output_dtype=np.float64)
sizes = [int(a) for a in sys.argv[1:]] or [16, 128, 340, 728, 1738, 4096, 8192]
reps = int(os.environ.get("REPS", 9))
for dt in (np.float64, np.float32, np.uint8):
    for n in sizes:
        x = synth.make_subimages(n, 128, dtype=dt)
        f.execute(x[:64], n_cols=20)
        f.execute(x, n_cols=20)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f.execute(x, n_cols=20)
            ts.append(time.perf_counter() - t0)
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        floor = max(x.nbytes / (HOST_READ_GBPS * 1e9), n * 16384 / (PCIE_GBPS * 1e9))
        print("host path  %-8s N=%5d   best %7.3f ms  median %7.3f ms  %9.0f sub-images/s  caller bytes %6.1f GB/s  wire %5.1f GB/s   ceiling (host read %g, PCIe %g GB/s): %.3f ms -> %.0f %% of it"
              % (np.dtype(dt).name, n, best * 1e3, med * 1e3, n / best, x.nbytes / best / 1e9, n * 16384 / best / 1e9, HOST_READ_GBPS, PCIE_GBPS,
                 floor * 1e3, 100 * floor / best), flush=True)
f.close()
