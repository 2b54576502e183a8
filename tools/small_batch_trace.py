"""Run 50 device-resident calls at N = 728 (for rocprofv3 --kernel-trace: kernel durations and the gaps between them)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 728
blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
x = torch.from_numpy(synth.make_subimages(n, 128, dtype=np.uint8)).to(dev)
y = torch.empty((n, 60), dtype=torch.float32, device=dev)
flow.reserve(2048)
for _ in range(50):
    flow.execute_device(x.data_ptr(), np.dtype(np.uint8), n, x.shape[1], y.data_ptr(), np.float32, 60, 60, stream=stream.cuda_stream)
torch.cuda.synchronize()
