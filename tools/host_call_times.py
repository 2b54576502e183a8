"""Wall time of one HOST-array call (the call the reference makes: numpy in, numpy out) of U11L-128 over a range of batch sizes and input
types, steady state: python tools/host_call_times.py [rows ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
x8 = synth.make_subimages(1024, 128, dtype=np.uint8)
for dt in (np.float64, np.float32, np.uint8):
    xs = x8.astype(dt)
    line = []
    for n in [int(a) for a in sys.argv[1:]] or (1, 18, 44, 130, 348, 728, 1024):
        x = xs[:n]
        for _ in range(30):
            flow.execute(x, n_cols=20)
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            flow.execute(x, n_cols=20)
        line.append("%d: %.1f" % (n, (time.perf_counter() - t0) / reps * 1e6))
    print("%-8s us per host call  " % np.dtype(dt).name + "  ".join(line), flush=True)
