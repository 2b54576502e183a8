"""Per-stage times of a product-expansion hierarchy (tests/helpers.product_hier_net) on the fused and the generic plan."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd.flow import Flow
from tests import helpers

side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 4096
nodes = helpers.product_hier_net(1, side=side)
x = np.random.default_rng(0).normal(size=(n, nodes[0].input_dim)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
yd = torch.empty((n, nodes[-1].output_dim), dtype=torch.float32, device="cuda")
for force in (False, True):
    flow = Flow(nodes, output_dtype=np.float32, force_generic=force)
    flow.reserve(n)
    for _ in range(3):
        flow.execute_device(xd.data_ptr(), np.float32, n, x.shape[1], yd.data_ptr(), np.float32, yd.shape[1], yd.shape[1])
    torch.cuda.synchronize()
    acc = None
    for _ in range(5):
        from pyfaceanalysis_amd import _capi
        _capi.check(_capi.lib().hg_flow_reset_profile(flow._handle().h))
        flow.execute_device(xd.data_ptr(), np.float32, n, x.shape[1], yd.data_ptr(), np.float32, yd.shape[1], yd.shape[1], profile=True)
        torch.cuda.synchronize()
        st = flow.stage_times()
        v = np.array([ms for _, ms, _ in st])
        acc = v if acc is None else np.minimum(acc, v)
    print("side %d %s: total %.3f ms, flops/row %d" % (side, "generic" if force else "fused", acc.sum(), flow.info().flops_per_row))
    if not force:
        for (nm, _, _), ms in zip(st, acc):
            print("   %.4f  %s" % (ms, nm[:150]))
    flow.close()
