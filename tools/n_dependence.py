"""Does the result of a row depend on the batch it arrives in?  (tests/test_gpu_host_path.py)"""
import numpy as np
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, _ = synth.cached_preset_blob("U11L-128")
x = synth.make_subimages(4096, 128, dtype=np.uint8)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
big = flow.execute(x)
for n in (1, 7, 16, 17, 32, 100, 128, 129, 340, 728, 1738, 2048, 4095):
    small = flow.execute(x[:n])
    d = float(np.abs(small.astype(np.float64) - big[:n]).max() / np.abs(big).max())
    print("N=%5d  max|diff|/max|y| = %.3g  identical=%s" % (n, d, np.array_equal(small, big[:n])))
