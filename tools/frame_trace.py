"""The configs[2] frame leg of bench.py (synthetic 17-stage cascade, four flows, one 1080p frame, 40 + 20 frames) on its own, for
rocprofv3 --kernel-trace: where a frame's time goes (tools/run_frame_trace.sh, tools/trace_frame_summary.py)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob(bench.PRESET)
dev = torch.device("cuda", 0)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
res = bench._frame_leg(flow, dev, 20, None)
print({k: res[k] for k in ("ms_per_frame", "rows_executed", "survivors_per_stage")})
