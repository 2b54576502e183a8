"""EXPERIMENTAL split-bf16 plan (HIGSFA_BF16X3=1, hg_fused_b3.hip) against the default exact-fp32 plan on U11L-128:
error of both against the float64 oracle, per-stage times, ms per 4096-row step.  A labelled side experiment."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mdp_restate
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
rows = 4096
xh = synth.make_subimages(rows, 128, dtype=np.float32)
x = torch.from_numpy(xh).to(dev)
ref = mdp_restate.execute_flow(nodes, xh[:128].astype(np.float64))
res = {}
for name, env in (("fp32", "0"), ("bf16x3", "1")):
    os.environ["HIGSFA_BF16X3"] = env
    flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
    flow.reserve(rows)
    y = torch.empty((rows, 60), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev)
    run = lambda prof=False: flow.execute_device(x.data_ptr(), np.float32, rows, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=st.cuda_stream, profile=prof)
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 500 * 1e3
    run(True)
    torch.cuda.synchronize()
    stages = [round(m * 1e3, 1) for _, m, _ in flow.stage_times()]
    yh = y.cpu().numpy()
    err = np.abs(yh[:128] - ref).max() / np.abs(ref).max()
    err20 = (np.abs(yh[:128, :20] - ref[:, :20]).max(axis=0) / np.abs(ref[:, :20]).max(axis=0)).max()
    res[name] = yh
    print("%-7s %.4f ms/step  max rel err vs float64 oracle %.2e (worst of the first 20 columns %.2e)  stage us %s" % (name, ms, err, err20, stages), flush=True)
    flow.close()
print("bf16x3 vs fp32 plan: max |d| / max|y| = %.2e" % (np.abs(res["bf16x3"] - res["fp32"]).max() / np.abs(res["fp32"]).max()))
