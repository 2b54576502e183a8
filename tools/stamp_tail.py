"""In-kernel timeline of the short-batch launches (diagnostic build: bash tools/build_diag_lib.sh;  HIGSFA_LIB=tools/ab/libhigsfa_diag.so
HIGSFA_STAMP=<stage> python tools/stamp_tail.py [rows]): stage 5 = the sub-tree launch (layers 5-7), 9 / 8 = the top-of-hierarchy launch of a
short / long batch.  Prints the stamped launch's report (stderr of the library) for the last of a few calls."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
blob, nodes = synth.cached_preset_blob("U11L-128")
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
x = synth.make_subimages(n, 128, dtype=np.uint8)
for _ in range(5):
    flow.execute(x)
