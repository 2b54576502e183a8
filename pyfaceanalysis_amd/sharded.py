"""Row-sharded execution over the GPUs of one node.

Every sub-image is independent (no cross-row term in any node; the caller's compaction,
FaceDetectUpdated.py:739-759, is per-row masking), so rank r of W takes the contiguous row block
``[r*ceil(N/W), ...)``, weights are replicated, and there is no exchange inside the 11 layers.
The only collective is the one the caller needs: an all-gather of the first k slow features
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_rows, world, rank):
    """Contiguous row block of ``rank``: equal blocks of ceil(n/world) rows (the last may be short
    or empty)."""
    per = (n_rows + world - 1) // world
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows), per


def gather_features(y_local, y_all):
    """All-gather equal-sized per-rank feature blocks (torch tensors) into ``y_all``
    ((world*rows, k), preallocated).  One collective per step; payload rows*k*4 bytes per rank
    (327 680 B at 4096 x 20 fp32) — latency-bound, so it is issued as a single all-gather."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(y_all, y_local)
    return y_all


class ShardedFlow(object):
    """``execute(x)`` over all ranks of the default process group: every rank passes the same
    global ``x`` (or only its shard with ``x_is_local=True``) and receives the full (N, k) result.

    ``execute_local`` is the per-rank compute callable ``(x_block ndarray) -> (rows, k) ndarray``;
    in production it is ``Flow.execute`` bound to this rank's GPU.  It is injected so that the
    sharding/gather logic can be exercised on CPU ranks (gloo) with any callable.
    """

    def __init__(self, execute_local, n_cols, device=None):
        self.execute_local = execute_local
        self.n_cols = int(n_cols)
        self.device = device

    def execute(self, x, x_is_local=False, n_total=None):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        if x_is_local:
            if n_total is None:
                raise ValueError("n_total is required with x_is_local=True")
            lo, hi, per = shard_bounds(n_total, world, rank)
            xb = x
            if xb.shape[0] != hi - lo:
                raise ValueError("rank %d: local block has %d rows, expected %d" % (rank, xb.shape[0], hi - lo))
        else:
            n_total = x.shape[0]
            lo, hi, per = shard_bounds(n_total, world, rank)
            xb = x[lo:hi]
        yb = np.zeros((per, self.n_cols), dtype=np.float32)
        if hi > lo:
            yb[:hi - lo] = np.asarray(self.execute_local(xb))[:, :self.n_cols]
        dev = self.device if self.device is not None else "cpu"
        y_local = torch.from_numpy(yb).to(dev)
        y_all = torch.empty((per * world, self.n_cols), dtype=torch.float32, device=dev)
        gather_features(y_local, y_all)
        return y_all[:n_total].cpu().numpy()
