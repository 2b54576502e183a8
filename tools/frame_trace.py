"""Run the synthetic 17-stage cascade on one 1080p frame 20 times (for rocprofv3 --kernel-trace: where a frame's time goes)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import grid, synth, synth_cascade
from pyfaceanalysis_amd.cascade import DeviceCascade, frame_windows
from pyfaceanalysis_amd.flow import Flow

blob, nodes = synth.cached_preset_blob("U11L-128")
dev = torch.device("cuda", 0)
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
rng = np.random.default_rng(synth.INPUT_SEED)
frame = torch.from_numpy(np.rint(synth._box3(rng.integers(0, 256, (1080, 1920), dtype=np.uint8))).astype(np.uint8)).to(dev)
pipe = dict(grid.FACE_PIPELINE)
boot = DeviceCascade([synth_cascade.Stage("Disc1", flow, synth_cascade.quantile_classifier(rng.normal(size=(50, 20)), 9, [0.0, 1.0]))], (128, 128), 20, pipe)
small = boot.prescale(frame)
boxes, level = frame_windows(int(small.shape[1]), int(small.shape[0]), 0.1, pipe, (128, 128))
subs = boot.patcher.extract(small.cpu().numpy(), boxes, (128, 128), dtype=np.uint8)
feats = flow.execute(subs, n_cols=20)
dc = DeviceCascade(synth_cascade.build_face_cascade(flow, feats, pipe, keep_fraction=0.1), (128, 128), 20, pipe)
for _ in range(20):
    out = dc.detect(dc.prescale(frame), smallest_face=0.1, windows=(boxes, level))
torch.cuda.synchronize()
print(out["counts"])
