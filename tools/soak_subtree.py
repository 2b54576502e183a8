"""Randomised soak of the sub-tree launches (k_subtree): calls of 1 .. 1100 rows, uint8 / float32 rows, against a flow planned with
HIGSFA_SUBTREE=0 (per-layer launches), bit for bit: python tools/soak_subtree.py <seconds>   (round 5: 33 439 calls in 90 s, 0 mismatches)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
blob, nodes = synth.cached_preset_blob("U11L-128")
x8 = synth.make_subimages(2048, 128, dtype=np.uint8)
os.environ["HIGSFA_SUBTREE"] = "0"
ref_flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
ref = ref_flow.execute(x8, n_cols=20)
del os.environ["HIGSFA_SUBTREE"]
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
assert "sub-trees in ONE launch" in flow.describe() and "sub-trees in ONE launch" not in ref_flow.describe()
rng = np.random.default_rng(11)
t0 = time.perf_counter(); calls = bad = 0
sizes = {}
while time.perf_counter() - t0 < float(sys.argv[1]):
    n = int(rng.integers(1, 1101)); off = int(rng.integers(0, 2048 - n + 1))
    dt = [np.uint8, np.float32][int(rng.integers(0, 2))]
    y = flow.execute(x8[off:off + n].astype(dt), n_cols=20)
    if not np.array_equal(y, ref[off:off + n]):
        bad += 1; sizes[n] = sizes.get(n, 0) + 1
    calls += 1
print("sub-tree soak: %d calls, mismatching calls %d %r" % (calls, bad, sizes))
