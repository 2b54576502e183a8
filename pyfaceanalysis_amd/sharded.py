"""Row-sharded execution over the GPUs of one node, one process per GPU.

Every sub-image is independent (no cross-row term in any node; the caller's compaction,
FaceDetectUpdated.py:739-759, is per-row masking), so rank r of W takes the contiguous row block
``[r*ceil(N/W), ...)``, weights are replicated, and there is no exchange inside the 11 layers.
The only collective is the one the caller needs: an all-gather of the first k slow features
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

``ShardedFlow`` is device-resident: the rank's block and the gathered features are torch tensors on
the rank's device, nothing passes through numpy or the host.  ``bench.py`` times exactly
``ShardedFlow.step``; ``tests/test_sharded_gloo.py`` runs the same class at world size 2 over gloo with
a CPU compute callable.  (A single process driving several GPUs uses the C entry
``hg_flow_execute_sharded`` instead — host buffers in and out, no collective.)
"""
from __future__ import annotations


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


_SIDE_STREAMS = {}


def _side_stream(torch, device):
    """ONE side stream per device and process, shared by every ShardedFlow on it.  Streams compete for the device's few
    hardware queues and which queue a new stream lands on depends on how many were created before it: measured at world size
    1 (profiles/r04_rccl_world1.txt), the step costs 0.522 ms with the process's first side stream, 0.544 with its second,
    and 0.908 with the fourth high-priority one, against 0.508 collective-free — so the choice is made once."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]


class ShardedFlow(object):
    """One rank's share of a sharded ``flow.execute``.

    ``execute_local(x_block, y_out, stream)`` is the per-rank compute: it reads the (m, input_dim) tensor
    ``x_block`` and writes the first ``n_cols`` features of every row into the (m, n_cols) float32 tensor
    ``y_out``, enqueued on the raw stream handle ``stream`` (0 on CPU).  In production it is
    ``Flow.execute_device`` on this rank's GPU (``ShardedFlow.for_flow``); the CPU tests inject a callable.

    ``rows`` is the block size ceil(N / world) every rank allocates for; a rank may pass fewer rows
    (the last block of a ragged batch): the rest of its block is zero in what the gather publishes
    (the two feature buffers are reused, so ``step`` clears whatever an earlier, fuller step left
    behind a short block).
    ``collective``: issue the all-gather (default: whenever a process group is initialised).

    On a GPU the gather of step i runs on a side stream under the kernels of step i+1 (two feature
    buffers, events in both directions); ``step`` returns the tensor the gather writes, valid once
    ``wait()`` has returned or ``done_event(i)`` of that step has been waited for, and ONLY until step
    i + 2 is enqueued, which writes the same buffer again: a consumer that needs it longer copies it.

    ``gather_stream``: "side" (default) as described above; "same": the collective is enqueued behind the kernels on their own
    stream — no hand-off between two queues, but the gather's time is no longer hidden under the next step's kernels.

    ``light_events``: scope of the event that hands a step's features from the kernels to the collective.  False (the
    default, also ``HIGSFA_GATHER_LIGHT_EVENTS=0`` / unset): an ordinary event, whose record publishes the kernels' writes
    to system scope — what any transport of the collective may rely on.  True: a device-scope event
    (``hipEventDisableSystemFence``; a default event's record held the next launch up by 13 us on an MI355X, 3-4 us more
    per step than this form) — enough when the collective reads the features with a kernel of THIS device, which is what
    RCCL's all-gather does, but until a run on more than one GPU has confirmed it the caller must ask for it: ``bench.py``
    does, after ``verify_against_blocking_gather`` has passed on every rank.  The two events in the other direction
    ("the gather has finished READING buffer b": an order, no data) are always device-scope.
    """

    def __init__(self, execute_local, n_cols, rows, device=None, collective=None, light_events=None, gather_stream="side"):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.execute_local = execute_local
        self.n_cols, self.rows = int(n_cols), int(rows)
        self.device = torch.device("cpu") if device is None else torch.device(device)
        self.cuda = self.device.type == "cuda"
        have_pg = dist.is_available() and dist.is_initialized()
        self.collective = have_pg if collective is None else bool(collective)
        if self.collective and not have_pg:
            raise RuntimeError("ShardedFlow: collective requested but no process group is initialised")
        self.world = dist.get_world_size() if self.collective else 1
        self.rank = dist.get_rank() if self.collective else 0
        mk = lambda r: torch.zeros((r, self.n_cols), dtype=torch.float32, device=self.device)
        self.ys = [mk(self.rows), mk(self.rows)]
        self.y_alls = [mk(self.rows * self.world), mk(self.rows * self.world)] if self.collective else None
        self._n = 0
        self._filled = [0, 0]          # rows of ys[b] that may hold features of an earlier step
        if gather_stream not in ("side", "same"):
            raise ValueError("gather_stream must be 'side' or 'same'")
        self.gather_stream = gather_stream
        if self.cuda:
            self.stream = torch.cuda.current_stream(self.device)
            self.comm = _side_stream(torch, self.device) if (self.collective and gather_stream == "side") else None
            self.gathered = [torch.cuda.Event(), torch.cuda.Event()]     # for the caller (done_event): recorded on the side stream
            self._light = None
            if self.collective and gather_stream == "side":
                # events ordering the two streams against each other: [0/1] "gather of buffer b enqueued so far is done"
                # (side stream -> kernels; a read-before-overwrite order, device scope), [2] "kernels of this step are done"
                # (kernels -> side stream; carries the features: scope chosen by light_events).  Created on self.device,
                # whatever device is current in the calling thread.
                import ctypes as C
                import os
                from . import _capi
                if light_events is None:
                    light_events = os.environ.get("HIGSFA_GATHER_LIGHT_EVENTS", "0") not in ("", "0")
                self.light_events = bool(light_events)
                self._capi, self._light = _capi, []
                dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
                for k in range(3):
                    h = C.c_void_p()
                    _capi.check(_capi.lib().hg_event_create_on(C.byref(h), int(dev_index), 1 if (k < 2 or self.light_events) else 0))
                    self._light.append(h)
                self._recorded = [False, False]

    @classmethod
    def for_flow(cls, flow, n_cols, rows, device, collective=None, light_events=None, gather_stream="side"):
        """Bind ``flow.execute_device`` (pyfaceanalysis_amd.flow.Flow on this rank's GPU)."""
        import numpy as np
        import torch
        np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8}
        flow.reserve(rows)

        def run(x_block, y_out, stream):
            if x_block.stride(1) != 1 or y_out.stride() != (n_cols, 1):
                raise ValueError("ShardedFlow: x rows and the feature buffer must be contiguous")
            flow.execute_device(x_block.data_ptr(), np_dt[x_block.dtype], x_block.shape[0], x_block.stride(0),
                                y_out.data_ptr(), np.float32, n_cols, n_cols, stream=stream)
        return cls(run, n_cols, rows, device=device, collective=collective, light_events=light_events, gather_stream=gather_stream)

    def step(self, x_local):
        """Enqueue one pass over this rank's block; returns the (world*rows, n_cols) gathered features
        (or the (rows, n_cols) local ones without a collective) of THIS step."""
        torch = self.torch
        b = self._n & 1
        self._n += 1
        m = int(x_local.shape[0])
        if m > self.rows:
            raise ValueError("rank %d: local block has %d rows, more than the %d allocated" % (self.rank, m, self.rows))
        if self.cuda and torch.cuda.current_device() != self.device.index and self.device.index is not None:
            with torch.cuda.device(self.device):      # events and streams of self.device: make it current for the calls below
                return self._step(x_local, b, m)
        return self._step(x_local, b, m)

    def _step(self, x_local, b, m):
        torch = self.torch
        if m < self._filled[b]:         # a fuller step used this buffer before: its rows m.. must not be published again
            if self.cuda:
                with torch.cuda.stream(self.stream):
                    self.ys[b][m:self._filled[b]].zero_()
            else:
                self.ys[b][m:self._filled[b]].zero_()
        self._filled[b] = m
        if m:
            self.execute_local(x_local, self.ys[b][:m], self.stream.cuda_stream if self.cuda else 0)
        if not self.collective:
            return self.ys[b]
        if self.cuda and self.gather_stream == "same":
            # The collective on the kernels' OWN stream (round 5): no event, no second queue, nothing for the next
            # step's first kernel to wait for but the gather itself, which is then serial with the kernels instead of hidden
            # under the next step's (measured beside the side-stream form at world size 1: profiles/r05_rccl_world1.txt).
            with torch.cuda.stream(self.stream):
                gather_features(self.ys[b], self.y_alls[b])
                self.gathered[b].record(self.stream)
            return self.y_alls[b]
        if self.cuda:
            L = self._capi.lib()
            # The NEXT step writes ys[1 - b]; the gather that read it (last step's) must be done before.  It almost always is,
            # long ago: ask first (non-blocking).  If not, the wait goes HERE, right in front of the record below, not in front of
            # the next step's first launch: a wait attaches to the next command of the stream, and a kernel dispatch cannot carry
            # a dependency — the runtime then puts a barrier packet of its own in front of it (the round-3 trace: two barrier
            # packets between two steps' kernels, 13 us) — while the record's marker packet can.
            nb = 1 - b
            if self._recorded[nb]:
                q = L.hg_event_query(self._light[nb])
                if q < 0:
                    self._capi.check(q)
                if q == 0:
                    self._capi.check(L.hg_stream_wait_event(self.stream.cuda_stream, self._light[nb]))
            self._capi.check(L.hg_event_record(self._light[2], self.stream.cuda_stream))
            self._capi.check(L.hg_stream_wait_event(self.comm.cuda_stream, self._light[2]))
            with torch.cuda.stream(self.comm):
                gather_features(self.ys[b], self.y_alls[b])
                self._capi.check(L.hg_event_record(self._light[b], self.comm.cuda_stream))
                self._recorded[b] = True
                self.gathered[b].record(self.comm)
        else:
            gather_features(self.ys[b], self.y_alls[b])
        return self.y_alls[b]

    def verify_against_blocking_gather(self, x_blocks, steps=6):
        """Check the overlapped gather against an independent, fully synchronous one: ``steps`` steps over DIFFERENT input
        blocks (``x_blocks[i % len(x_blocks)]``: consecutive steps give different features, so a peer block that arrives one
        step stale, or that was read before its writes were visible, cannot pass), and after every step — once everything is
        complete — a blocking all-gather of this rank's features straight from the buffer the kernels wrote.  Every rank
        compares EVERY rank's block.  Returns True only if every step matched on every rank (a MIN all-reduce), so that
        all ranks take the same decision.  Not for timed regions."""
        import torch.distributed as dist
        torch = self.torch
        if not self.collective:
            return True
        ok = True
        for i in range(int(steps)):
            xb = x_blocks[i % len(x_blocks)]
            y_over = self.step(xb)
            b = (self._n - 1) & 1
            self.wait()
            ref = torch.zeros_like(y_over)
            dist.all_gather_into_tensor(ref, self.ys[b])
            if self.cuda:
                torch.cuda.synchronize(self.device)
            ok = ok and bool(torch.equal(ref, y_over))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if self.cuda:
            torch.cuda.synchronize(self.device)
        return bool(int(flag.item()) == 1)

    def done_event(self, step_index=None):
        """The event recorded after the gather of step ``step_index`` (default: the last one enqueued); only the
        two most recent steps have one.  ``None`` on CPU or without a collective (then the result of a step is
        ordered on ``self.stream`` itself)."""
        last = self._n - 1
        i = last if step_index is None else int(step_index)
        if not (last - 1 <= i <= last) or i < 0:
            raise ValueError("step %d: only the two most recent steps (%d, %d) still own a buffer" % (i, last - 1, last))
        return self.gathered[i & 1] if (self.cuda and self.collective) else None

    def close(self):
        if getattr(self, "_light", None):
            self.wait()
            for h in self._light:
                self._capi.lib().hg_event_destroy(h)
            self._light = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def wait(self):
        """Block until everything enqueued so far (kernels and gathers) is complete."""
        if self.cuda:
            self.torch.cuda.synchronize(self.device)

    def execute(self, x_local, n_total=None):
        """Synchronous form: the (n_total, n_cols) features of the whole batch, on this rank's device.
        ``x_local`` is this rank's block (``shard_bounds``); n_total defaults to rows * world."""
        y = self.step(x_local)
        self.wait()
        n_total = self.rows * self.world if n_total is None else int(n_total)
        lo, hi, per = shard_bounds(n_total, self.world, self.rank)
        if (self.world > 1 and per != self.rows) or per > self.rows:       # rank r's rows sit at r * self.rows in the gathered matrix
            raise ValueError("n_total %d does not match blocks of %d rows on %d ranks" % (n_total, self.rows, self.world))
        if int(x_local.shape[0]) != hi - lo:
            raise ValueError("rank %d: local block has %d rows, expected %d" % (self.rank, x_local.shape[0], hi - lo))
        return y[:n_total]
