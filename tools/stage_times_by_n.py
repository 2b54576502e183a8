"""Per-stage event times (us) of one device-resident call of U11L-128 for a range of batch sizes: python tools/stage_times_by_n.py [rows ...]
(events around every launch: each reads ~1 us long; the sum is not the call time — tools/call_times.py has that)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
flow.reserve(4096)
xs = torch.from_numpy(synth.make_subimages(4096, 128, dtype=np.uint8)).to(dev)
for n in [int(a) for a in sys.argv[1:]] or (728, 1024, 1400, 1738, 2048, 3000, 4096):
    x = xs[:n]
    y = torch.empty((n, 60), dtype=torch.float32, device=dev)
    for _ in range(30):
        flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    from pyfaceanalysis_amd import _capi
    _capi.check(_capi.lib().hg_flow_reset_profile(flow._handle(None).h))
    reps = 50
    for _ in range(reps):
        flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream, profile=True)
        torch.cuda.synchronize()
    t = flow.stage_times()
    print("N=%5d  " % n + "  ".join("%5.1f" % (ms * 1e3 / max(1, cnt)) for _, ms, cnt in t) + "   sum %.1f" % sum(ms * 1e3 / max(1, cnt) for _, ms, cnt in t), flush=True)
