"""BASELINE.json configs[2] (first cascade stage only): one synthetic 1920x1080 frame, smallest_face = 0.1,
prescaled to 1000x562 like the reference (FaceDetectUpdated.py:551-556), all 10 pyramid levels' windows
(1738) cut on the GPU and pushed through the 11-layer net; everything device resident.
The data-dependent later stages of the cascade (coordinate updates, discards) are host logic outside the
hot path and are not part of this figure."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfaceanalysis_amd import grid, synth
from pyfaceanalysis_amd.flow import Flow
from pyfaceanalysis_amd.patches import Patcher

side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
preset = "U11L-128" if side == 128 else "U11L-64"
blob, nodes = synth.cached_preset_blob(preset)
flow = Flow.from_blob(blob, output_dtype=np.float32)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(synth.INPUT_SEED)
frame = torch.from_numpy(np.rint(synth._box3(rng.integers(0, 256, (1080, 1920), dtype=np.uint8))).astype(np.uint8)).to(dev)
pw, ph = grid.prescaled_size(1920, 1080)
levels = grid.frame_boxes(pw, ph, 0.1, subimage_size=(side, side))
boxes = [torch.from_numpy(b).to(dev) for _, b in levels]
n_total = sum(len(b) for _, b in levels)
whole = torch.tensor([[0.0, 0.0, 1920.0, 1080.0]], dtype=torch.float64, device=dev)
small = torch.empty((ph, pw), dtype=torch.uint8, device=dev)
subs = [torch.empty((len(b), side * side), dtype=torch.uint8, device=dev) for _, b in levels]
feats = [torch.empty((len(b), 20), dtype=torch.float32, device=dev) for _, b in levels]
pt = Patcher()
flow.reserve(max(len(b) for _, b in levels))
st = torch.cuda.current_stream(dev).cuda_stream


def one_frame():
    # prescale = nearest resize = EXTENT over the whole frame (PIL resize NEAREST uses the same rule)
    pt.extract_device(frame.data_ptr(), np.uint8, 1080, 1920, 1920, whole.data_ptr(), 1, (pw, ph), small.data_ptr(), np.uint8, pw * ph, st)
    for i in range(len(levels)):
        n = subs[i].shape[0]
        pt.extract_device(small.data_ptr(), np.uint8, ph, pw, pw, boxes[i].data_ptr(), n, (side, side), subs[i].data_ptr(), np.uint8,
                          side * side, st)
        flow.execute_device(subs[i].data_ptr(), np.uint8, n, side * side, feats[i].data_ptr(), np.float32, 20, 20, stream=st)


all_boxes = torch.cat(boxes)
all_subs = torch.empty((n_total, side * side), dtype=torch.uint8, device=dev)
all_feats = torch.empty((n_total, 20), dtype=torch.float32, device=dev)
flow.reserve(n_total)


def one_frame_batched():
    # "all resolutions could be processed also at once" (FaceDetectUpdated.py:599): one extraction, one execute
    pt.extract_device(frame.data_ptr(), np.uint8, 1080, 1920, 1920, whole.data_ptr(), 1, (pw, ph), small.data_ptr(), np.uint8, pw * ph, st)
    pt.extract_device(small.data_ptr(), np.uint8, ph, pw, pw, all_boxes.data_ptr(), n_total, (side, side), all_subs.data_ptr(), np.uint8,
                      side * side, st)
    flow.execute_device(all_subs.data_ptr(), np.uint8, n_total, side * side, all_feats.data_ptr(), np.float32, 20, 20, stream=st)


for _ in range(3):
    one_frame()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    one_frame()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
one_frame_batched()
torch.cuda.synchronize()
# same features either way: a window's result does not depend on its batch beyond rounding (levels of a single
# batch tile take the unfused first-layer kernels, whose accumulation order differs in the last bits)
lvl = torch.cat(feats)
dev_max = float((lvl - all_feats).abs().max() / all_feats.abs().max())
assert dev_max <= 1e-5, dev_max
print("  level-by-level vs one batch: max relative difference %.1e" % dev_max)
t0 = time.perf_counter()
for _ in range(K):
    one_frame_batched()
torch.cuda.synchronize()
dtb = (time.perf_counter() - t0) / K
print("  all levels in one batch: %.2f ms/frame = %.1f frames/s = %.0f windows/s" % (dtb * 1e3, 1 / dtb, n_total / dtb))
print("frame 1920x1080 -> %dx%d, %d levels, %d windows of %dx%d: %.2f ms/frame = %.1f frames/s = %.0f windows/s (largest level %d windows)"
      % (pw, ph, len(levels), n_total, side, side, dt * 1e3, 1 / dt, n_total / dt, max(len(b) for _, b in levels)))
