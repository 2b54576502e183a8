import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
blob, nodes = synth.cached_preset_blob("U11L-128")
x = synth.make_subimages(4101, 128, dtype=np.uint8)
os.environ["HIGSFA_NO_MERGE"] = "1"
plain = Flow.from_blob(blob, device=0, output_dtype=np.float32)
del os.environ["HIGSFA_NO_MERGE"]
merged = Flow.from_blob(blob, device=0, output_dtype=np.float32)
ref = plain.execute(x, n_cols=20)
dev = torch.device("cuda", 0)
xa, xb = torch.from_numpy(x[:4096]).to(dev), torch.from_numpy(np.ascontiguousarray(x[5:4101])).to(dev)
ya, yb = (torch.empty((4096, 20), dtype=torch.float32, device=dev) for _ in range(2))
st = torch.cuda.current_stream(dev).cuda_stream
for mode in ("sync_every", "back_to_back"):
    bad = 0
    for rep in range(5):
        for i in range(60):
            src, dst = (xa, ya) if i % 2 == 0 else (xb, yb)
            merged.execute_device(src.data_ptr(), np.uint8, 4096, 16384, dst.data_ptr(), np.float32, 20, 20, stream=st)
            if mode == "sync_every":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        a, b = ya.cpu().numpy(), yb.cpu().numpy()
        ra = (a != ref[:4096]).any(axis=1); rb = (b != ref[5:4101]).any(axis=1)
        bad += int(ra.sum()) + int(rb.sum())
        if ra.any():
            w = np.nonzero(ra)[0]
            # does a wrong row equal the OTHER input's features at that position (a stale block)?
            stale = int(sum(np.array_equal(a[r], ref[r + 5]) for r in w[:200]))
            print(mode, "rep", rep, "wrong rows in ya:", len(w), "first", w[:8], "tiles", sorted(set((w // 16).tolist()))[:10], "equal to the other input's row:", stale, "of", min(len(w), 200))
    print(mode, "total wrong rows", bad)
