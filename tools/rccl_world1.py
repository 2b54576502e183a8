"""ShardedFlow.step with the collective branch on ONE GPU: RCCL initialised in-process at world size 1 (no launcher), the
same steps collective-free before and after.  Prints ms/step of the three; under `rocprofv3 --kernel-trace` the trace shows
where the difference sits (tools/kernel_positions.py does not apply: use the raw csv)."""
import os, socket, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
from pyfaceanalysis_amd.sharded import ShardedFlow

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
blob, nodes = synth.cached_preset_blob("U11L-128")
flow = Flow.from_blob(blob, device=0, output_dtype=np.float32)
dev = torch.device("cuda", 0)
rows = 4096
x = torch.from_numpy(synth.make_subimages(rows, 128, dtype=np.float32)).to(dev)
VARIANTS = {"free": ("collective_free", False, None, "side"), "ordinary": ("rccl_world1, ordinary hand-off event", True, False, "side"),
            "light": ("rccl_world1, device-scope hand-off event", True, True, "side"),
            "same": ("rccl_world1, gather on the kernels' own stream", True, None, "same")}
order = sys.argv[2].split(",") if len(sys.argv) > 2 else ["free", "ordinary", "light", "same", "free"]
for name, coll, light, gs in [VARIANTS[k] for k in order]:
    sf = ShardedFlow.for_flow(flow, 20, rows, dev, collective=coll, light_events=light, gather_stream=gs)
    for _ in range(300):
        sf.step(x)
    sf.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        sf.step(x)
    sf.wait()
    print("%-44s %.4f ms/step" % (name, (time.perf_counter() - t0) / steps * 1e3), flush=True)
dist.destroy_process_group()
