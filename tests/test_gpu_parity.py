"""GPU parity: HIP path (through the C ABI) vs the float64 oracle on identical batches.
Tolerance (BASELINE.json north_star): max|y - ref| / max|ref| <= 1e-4 in fp32."""
import numpy as np
import pytest

from oracle import mdp_restate as oracle
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_err(y, ref):
    return float(np.abs(np.asarray(y, dtype=np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("preset,kw", [
    ("T3L-8", {}), ("T5L-16", {}), ("T5L-16", {"layout": "separate"}), ("T5L-16", {"node_kind": "igsfa"}),
])
def test_small_nets_match_oracle(native_lib, nets, preset, kw, force_generic):
    nodes = nets(preset, **kw)
    side = synth.preset_input_side(preset)
    x = synth.make_subimages(67, side, dtype=np.float64)       # ragged: not a multiple of 16
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes, force_generic=force_generic)
    y = flow.execute(x)
    assert y.dtype == np.float64 and y.shape == ref.shape
    assert rel_err(y, ref) <= TOL
    flow.close()


@pytest.mark.parametrize("force_generic", [True, False])
def test_u11l_128_matches_oracle(native_lib, nets, force_generic):
    """BASELINE.json configs[0]: the 11-layer net on 256 sub-images of 128x128."""
    nodes = nets("U11L-128")
    x = synth.make_subimages(256, 128, dtype=np.float32)
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes, force_generic=force_generic)
    y = flow.execute(x)
    err = rel_err(y, ref)
    percol = np.abs(y - ref).max(axis=0)[:20] / np.abs(ref).max()
    print("U11L-128 generic=%s max|d|/max|ref| = %.3e; worst of first 20 cols %.3e" % (force_generic, err, percol.max()))
    assert err <= TOL
    flow.close()
