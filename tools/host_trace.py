"""One traced host call per (dtype, N): HIGSFA_HOST_TRACE=1 makes hg_flow_execute print its timeline (tickets, pieces, passes)."""
import os, sys
os.environ["HIGSFA_HOST_TRACE"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
blob, nodes = synth.cached_preset_blob("U11L-128")
f = Flow.from_blob(blob, output_dtype=np.float64)
for dt in (np.float64, np.uint8):
    for n in [int(a) for a in sys.argv[1:]] or [16, 728, 4096]:
        x = synth.make_subimages(n, 128, dtype=dt)
        for _ in range(4):
            f.execute(x, n_cols=20)
        sys.stderr.write("----\n")
f.close()
