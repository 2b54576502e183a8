import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import mdp_restate as oracle
from pyfaceanalysis_amd.flow import Flow
from tests import helpers
bad = 0; tails = 0; n = 0
for kind, maker, seeds in (("net", helpers.fuzz_net, range(0, 150)), ("prod", helpers.fuzz_product_net, range(0, 60)), ("igsfa", helpers.fuzz_igsfa_net, range(0, 40))):
    for seed in seeds:
        nodes = maker(seed)
        rng = np.random.default_rng(seed)
        x = rng.normal(size=(int(rng.integers(1, 70)), nodes[0].input_dim)) * 1.5
        f = Flow(nodes)
        d = f.describe()
        tails += "no unpack pass" in d
        y = f.execute(x)
        k = int(rng.integers(1, nodes[-1].output_dim + 1))
        yk = f.execute(x, n_cols=k)
        ref = oracle.execute_flow(nodes, x)
        err = np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)
        ok = err <= 1e-4 and np.array_equal(yk, y[:, :k])
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", kind, seed, err, f.info().plan_kind)
        f.close()
print("wide fuzz: %d flows, %d with the top-of-hierarchy launch, %d mismatches" % (n, tails, bad))
