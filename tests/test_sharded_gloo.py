"""CPU, world_size 2 over gloo: the row-sharding + all-gather logic of pyfaceanalysis_amd.sharded
 — the same ``ShardedFlow.step`` that bench.py times on the GPUs.  The per-rank compute callable is
injected; here it is the oracle, because the HIP path needs a GPU — the test checks the partition/gather, not the
arithmetic.  Success = exit code 0 of the launcher and one result file per rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pyfaceanalysis_amd.sharded import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle import mdp_restate
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.sharded import ShardedFlow, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nodes = synth.build_preset("T3L-8")
K = 4

def run(xb, y_out, stream):          # the role Flow.execute_device plays on a GPU
    y_out.copy_(torch.from_numpy(mdp_restate.execute_flow(nodes, xb.numpy())[:, :K].astype(np.float32)))

checked = 0
for n in (37, 1, 64, 2):
    x = synth.make_subimages(n, 8, seed=5, dtype=np.float64)
    ref = mdp_restate.execute_flow(nodes, x)[:, :K]
    lo, hi, per = shard_bounds(n, world, rank)
    sf = ShardedFlow(run, n_cols=K, rows=per)
    assert sf.collective and sf.world == world
    xl = torch.from_numpy(x[lo:hi])
    y = sf.execute(xl, n_total=n)
    assert tuple(y.shape) == (n, K) and np.abs(y.numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), (rank, n)
    # steps alternate between two buffers: three more steps give the same rows every time
    for _ in range(3):
        y2 = sf.step(xl)
        sf.wait()
        assert torch.equal(y2[:n], y)
    checked += 1
# inputs that change from step to step (a block that arrives one step stale cannot pass): every step's WHOLE gathered matrix
# against features computed independently for every rank's block, and the library's own check against a blocking gather
n = 48
lo, hi, per = shard_bounds(n, world, rank)
sf = ShardedFlow(run, n_cols=K, rows=per)
xs = [synth.make_subimages(n, 8, seed=50 + i, dtype=np.float64) for i in range(3)]
refs = [mdp_restate.execute_flow(nodes, x)[:, :K].astype(np.float32) for x in xs]
for i in range(7):
    y = sf.step(torch.from_numpy(xs[i % 3][lo:hi]))
    sf.wait()
    assert np.abs(y[:n].numpy() - refs[i % 3]).max() <= 1e-6 * np.abs(refs[i % 3]).max(), (rank, i)
assert sf.verify_against_blocking_gather([torch.from_numpy(x[lo:hi]) for x in xs], steps=5)
dist.barrier()
dist.destroy_process_group()
with open(os.path.join({out!r}, "rank%d.ok" % rank), "w") as fh:
    fh.write("%d\n" % checked)
"""


def test_shard_bounds():
    assert shard_bounds(32768, 8, 3) == (12288, 16384, 4096)
    assert shard_bounds(10, 4, 3) == (9, 10, 3)
    assert shard_bounds(2, 4, 3) == (2, 2, 1)
    assert shard_bounds(0, 2, 1) == (0, 0, 0)
    covered = []
    for r in range(5):
        lo, hi, _ = shard_bounds(37, 5, r)
        covered += list(range(lo, hi))
    assert covered == list(range(37))


def test_single_process_without_collective():
    """No process group: ShardedFlow degrades to the local pass (what `python bench.py` runs at N = 1)."""
    import torch
    from pyfaceanalysis_amd.sharded import ShardedFlow
    calls = []

    def run(xb, y_out, stream):
        calls.append(int(xb.shape[0]))
        y_out.copy_(xb[:, :3].float() * 2)
    sf = ShardedFlow(run, n_cols=3, rows=8)
    assert not sf.collective and sf.world == 1
    x = torch.arange(40, dtype=torch.float64).reshape(8, 5)
    assert torch.equal(sf.execute(x), x[:, :3].float() * 2)
    assert torch.equal(sf.execute(x[:5], n_total=5), x[:5, :3].float() * 2) and calls == [8, 5]
    with pytest.raises(ValueError):
        sf.step(torch.zeros(9, 5))


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    # --standalone: the launcher picks a free rendezvous port itself (no port race to retry around)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=2", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    for rank in range(2):
        f = tmp_path / ("rank%d.ok" % rank)
        assert f.exists() and f.read_text().strip() == "4", out[-3000:]


def test_bench_starts_its_own_ranks_or_says_why_not():
    """`python bench.py --gpus N` (N > 1) without a launcher is a parent that never touches HIP: it checks the visible
    devices and starts torch.distributed.run as a child.  This container has no GPU, so the parent must stop with exit code 2
    and a message — not fall into the single-rank path, not hang, not raise."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box would really start the ranks")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 2, (r.returncode, r.stderr.decode(errors="replace")[-2000:])
    assert b"--gpus 2 asked for" in r.stderr and not r.stdout.strip()
