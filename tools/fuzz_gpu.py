"""One-off wide fuzz on the GPU: helpers.fuzz_net seeds [a, b) (append "igsfa" for helpers.fuzz_igsfa_net) through the fused plan vs the float64 oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import mdp_restate as oracle
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

a, b = int(sys.argv[1]), int(sys.argv[2])
maker = helpers.fuzz_igsfa_net if len(sys.argv) > 3 and sys.argv[3] == "igsfa" else helpers.fuzz_net
worst, bad = 0.0, []
for seed in range(a, b):
    nodes = maker(seed)
    n = [1, 15, 16, 17, 33, 100, 257][seed % 7]
    x = np.random.default_rng(seed).normal(size=(n, nodes[0].input_dim)) * 1.5
    ref = oracle.execute_flow(nodes, x)
    flow = Flow(nodes)
    kind = flow.info().plan_kind
    err = float(np.abs(flow.execute(x) - ref).max() / np.abs(ref).max())
    flow.close()
    worst = max(worst, err)
    if err > 1e-4 or kind != 1:
        bad.append((seed, kind, err))
print("seeds %d..%d: worst rel err %.2e, failures/non-fused: %s" % (a, b - 1, worst, bad))
