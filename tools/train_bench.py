"""BASELINE.json configs[4]: one SFA training step (per-node covariance accumulation + generalised
eigendecomposition) on 100k synthetic 128x128 patches, layer-0 geometry (1024 nodes x 16 pixels)."""
import os, sys, time
import numpy as np
import torch                                  # before libhigsfa: one ROCm runtime per process (see _capi.lib)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.linalg
from pyfaceanalysis_amd import nodes as N
from pyfaceanalysis_amd.train import sfa_train_layer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
side = 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(12345600)
# slowly varying sequence: exponentially smoothed box-filtered noise, quantised to 8 bits
x = torch.empty((n, side * side), dtype=torch.uint8, device=dev)
state = torch.rand((1, 1, side, side), device=dev, generator=g) * 255
for i0 in range(0, n, 2000):
    m = min(2000, n - i0)
    noise = torch.rand((m, 1, side, side), device=dev, generator=g) * 255
    noise = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(noise, (1, 1, 1, 1), mode="replicate"), 3, 1)
    out = torch.empty_like(noise)
    s = state
    for t in range(m):                       # first-order recursion along time
        s = 0.9 * s + 0.1 * noise[t:t + 1]
        out[t] = s[0]
    state = s
    x[i0:i0 + m] = out.round().clamp(0, 255).to(torch.uint8).reshape(m, -1)
torch.cuda.synchronize()
conn = N.Rectangular2dSwitchboard((side, side), (4, 4), (4, 4), 1).connections.reshape(-1, 16)
sfa_train_layer(x.data_ptr(), conn, x_dtype=np.uint8, n=1000, ldx=side * side)        # warm-up (library handles)
t0 = time.perf_counter()
evals, evecs, mean, tms = sfa_train_layer(x.data_ptr(), conn, x_dtype=np.uint8, n=n, ldx=side * side)
wall = time.perf_counter() - t0
# check a few nodes against scipy in float64
worst_val = worst_vec = 0.0
for k in (0, 511, 1023):
    xk = x[:, torch.from_numpy(conn[k].astype(np.int64)).to(dev)].double().cpu().numpy()
    B = np.cov(xk.T); dx = xk[1:] - xk[:-1]; A = dx.T @ dx / (n - 1)
    w, v = scipy.linalg.eigh(A, B)
    worst_val = max(worst_val, float(np.abs(evals[k] / w - 1).max()))
    worst_vec = max(worst_vec, float(np.abs(np.diag(np.abs(v.T @ B @ evecs[k])) - 1).max()))
flops = n * 1024.0 * (16 * 16 * 2) * 2
print("SFA train step: %d patches of 128x128 (uint8, %.2f GB), 1024 nodes x 16: statistics %.1f ms (%.1f GFLOP/s fp64, %.0f GB/s of input), "
      "generalized eigen-solve %.2f ms, wall %.1f ms; eigenvalues vs scipy %.1e, eigenvectors %.1e (budget 1e-5)"
      % (n, n * side * side / 1e9, tms[0], flops / tms[0] / 1e6, n * side * side / tms[0] / 1e6, tms[1], wall * 1e3, worst_val, worst_vec))
