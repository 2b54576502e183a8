"""CPU, world_size 2 over gloo: the row-sharding + all-gather logic of pyfaceanalysis_amd.sharded
(SURVEY.md §8e).  The per-rank compute callable is injected; here it is the oracle, because the
HIP path needs a GPU — the test checks the partition/gather, not the arithmetic."""
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
import numpy as np, torch.distributed as dist
from oracle import mdp_restate
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.sharded import ShardedFlow, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nodes = synth.build_preset("T3L-8")
for n in (37, 1, 64):
    x = synth.make_subimages(n, 8, seed=5, dtype=np.float64)
    ref = mdp_restate.execute_flow(nodes, x)[:, :4]
    sf = ShardedFlow(lambda xb: mdp_restate.execute_flow(nodes, xb), n_cols=4)
    y = sf.execute(x)
    assert y.shape == (n, 4) and np.abs(y - ref).max() <= 1e-6 * np.abs(ref).max(), (rank, n)
    lo, hi, per = shard_bounds(n, world, rank)
    y2 = sf.execute(x[lo:hi], x_is_local=True, n_total=n)
    assert np.array_equal(y, y2)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
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


def test_world_size_2_gloo(tmp_path):
    import socket
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = ""
    for attempt in range(3):                          # rendezvous can lose a port race on a busy box: retry on a fresh port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode(errors="replace")
        if r.returncode == 0 and "rank 0 ok" in out and "rank 1 ok" in out:
            return
        if "AssertionError" in out:                   # a real numerical / logic failure: do not retry
            break
    raise AssertionError(out[-3000:])
