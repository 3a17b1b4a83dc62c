"""Batch sharding of image-pair inference over the GPUs of one node (one process per GPU).

The reference has no multi-GPU code at all; PWCDCNet has no cross-sample
operation (convs + LeakyReLU only, models/PWCNet.py:26-36), so the path shards over the batch with
ZERO communication inside a forward.  The only exchanges are
  * one broadcast of the parameters from rank 0 at start-up (37.5 MB fp32), and
  * a gather of the flow fields to rank 0 per batch (229 KB per 1024x448 pair),
both over RCCL/xGMI when the backend is "nccl" (that IS RCCL on ROCm); "gloo" is used by the CPU
tests.  No all-reduce exists on this path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n_items over world_size ranks; the first (n % world) ranks get one more."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world %d" % (rank, world_size))
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Broadcast all parameters/buffers from `src` as ONE flat buffer; returns the bytes sent."""
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if flat.is_cuda and dist.get_backend(group) == "gloo":      # CPU rehearsal of the RCCL path
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    if hasattr(module, "invalidate_plans"):
        module.invalidate_plans()
    return flat.numel() * flat.element_size()


class FlowGather:
    """Gather of per-rank flow fields [b_r, ...] to `dst` with every buffer allocated ONCE: the padded send buffer of a
    short rank and, on `dst`, one [world * max(counts), ...] receive buffer whose per-rank chunks are the collective's
    gather list (no per-step allocation or torch.cat on the timed path when all ranks hold the same count; a ragged
    step compacts into a second preallocated buffer).  The returned tensor is reused by the next call."""

    def __init__(self, counts: List[int], tail: Tuple[int, ...], dtype: torch.dtype, device: torch.device, dst: int = 0, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if len(counts) != self.world:
            raise ValueError("counts must have one entry per rank")
        self.counts, self.tail, self.dst, self.group = list(counts), tuple(tail), dst, group
        self.m = max(counts) if counts else 0
        self.device = torch.device(device)
        # gloo has no device gather: the CPU rehearsal of the RCCL path stages through host buffers
        self.host = self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        wire = torch.device("cpu") if self.host else self.device
        self.send = torch.zeros((self.m,) + self.tail, dtype=dtype, device=wire) if (counts[self.rank] < self.m or self.host) else None
        self.recv = self.out = None
        if self.rank == dst and self.m:
            self.recv = torch.empty((self.world * self.m,) + self.tail, dtype=dtype, device=wire)
            self.chunks = list(self.recv.split(self.m, 0))
            ragged = any(c != self.m for c in counts)
            if ragged or self.host:
                self.out = torch.empty((sum(counts),) + self.tail, dtype=dtype, device=self.device)

    def __call__(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        c = self.counts[self.rank]
        if local.shape[0] != c or tuple(local.shape[1:]) != self.tail:
            raise ValueError("rank %d holds %s, FlowGather was built for [%d,%s]" % (self.rank, tuple(local.shape), c, ",".join(map(str, self.tail))))
        if self.m == 0:
            return local if self.rank == self.dst else None
        if self.send is not None:
            self.send[:c].copy_(local)
            send = self.send
        else:
            send = local.contiguous()
        if self.rank != self.dst:
            dist.gather(send, gather_list=None, dst=self.dst, group=self.group)
            return None
        dist.gather(send, gather_list=self.chunks, dst=self.dst, group=self.group)
        if self.out is None:
            return self.recv                                   # equal counts: rank order == batch order, no copy
        off = 0
        for chunk, n in zip(self.chunks, self.counts):
            self.out[off:off + n].copy_(chunk[:n])
            off += n
        return self.out


class AsyncFlowGather:
    """FlowGather off the compute stream: step k's flows are gathered while forward k+1 runs.

    bench.py's N > 1 loop used to enqueue the gather to rank 0 behind each forward on the compute stream, so forward k+1 waited for
    gather k (26 MB into rank 0 at N = 8, an estimated 0.2-0.3 ms of a 10.7 ms step).  Here the flows of step k are copied into one of
    two staging buffers on a SIDE stream (which waits for an event recorded behind forward k), the collective is issued on that
    stream (RCCL orders itself after the stream that is current when it is called), and an event marks its end; the compute stream
    waits for the staging copy only (microseconds), never for the collective.  ``submit`` returns at once; ``result(k)`` gives step k's gathered tensor on ``dst`` (None elsewhere)
    after making the CALLER's stream wait for its event -- valid until the submit two steps later reuses the buffers.  With a CPU
    / gloo group there are no streams and the call degenerates to the synchronous gather: same results, which is what the
    world-size-2 test pins; the overlap itself is checked on the GPU with event timestamps (tests/test_gpu_rccl.py)."""

    def __init__(self, counts: List[int], tail: Tuple[int, ...], dtype: torch.dtype, device: torch.device, dst: int = 0, group=None):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.rank = dist.get_rank(group)
        self.dst = dst
        self.count = counts[self.rank]
        # two independent gathers: their receive buffers are the double buffer of the results
        self.g = [FlowGather(counts, tail, dtype, device, dst, group) for _ in range(2)]
        self.stage = [torch.empty((self.count,) + tuple(tail), dtype=dtype, device=self.device) for _ in range(2)] if self.cuda else None
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.ready = [torch.cuda.Event() for _ in range(2)] if self.cuda else None          # recorded behind the forward
        self.done = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if self.cuda else None   # recorded behind the gather
        self.copied = [torch.cuda.Event() for _ in range(2)] if self.cuda else None         # recorded behind the copy into staging
        self.out = [None, None]
        self.k = 0

    def submit(self, local: torch.Tensor) -> int:
        """enqueue the gather of this step's flows; returns the step number to pass to result()"""
        k, i = self.k, self.k & 1
        self.k += 1
        if not self.cuda:
            self.out[i] = self.g[i](local)
            return k
        cur = torch.cuda.current_stream(self.device)
        self.ready[i].record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready[i])
            self.stage[i].copy_(local, non_blocking=True)
            self.copied[i].record(self.side)
            local.record_stream(self.side)
            self.out[i] = self.g[i](self.stage[i])
            self.done[i].record(self.side)
        # The caller's stream waits for the COPY (a few microseconds of device-to-device traffic that start the moment the forward
        # ends), not for the collective: `local` may be a buffer the next forward writes again (PWCDCNet(borrow_output=True) hands out
        # the plan's own flow buffer), and without this the copy would merely be very likely to win that race.
        cur.wait_event(self.copied[i])
        return k

    def result(self, k: int) -> Optional[torch.Tensor]:
        if k < self.k - 2 or k >= self.k:
            raise ValueError("step %d is not one of the two most recent submits (%d submitted)" % (k, self.k))
        i = k & 1
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.done[i])
        return self.out[i]

    def synchronize(self) -> None:
        if self.cuda:
            self.side.synchronize()


_GATHERS = {}


def gather_flows(local: torch.Tensor, counts: List[int], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather per-rank flow fields [b_r,2,h,w] to `dst`; returns the [sum b_r,2,h,w] batch there, None elsewhere.

    Ranks may hold different counts (ragged tail); shorter shards are padded to the longest for the collective and
    trimmed on arrival.  Buffers are allocated once per (counts, shape, dtype, device) and reused (FlowGather): the
    result is overwritten by the next call with the same geometry."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    if local.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d items, counts says %d" % (rank, local.shape[0], counts[rank]))
    key = (tuple(counts), tuple(local.shape[1:]), local.dtype, str(local.device), dst, id(group))
    g = _GATHERS.get(key)
    if g is None:
        if len(_GATHERS) > 16:
            _GATHERS.clear()
        g = _GATHERS[key] = FlowGather(counts, tuple(local.shape[1:]), local.dtype, local.device, dst, group)
    return g(local)


class ShardedFlow:
    """Run a per-rank flow function over this rank's slice of a global batch and gather on rank 0."""

    def __init__(self, forward_fn, group=None):
        self.forward_fn = forward_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, global_batch_size: int, load_local) -> Optional[torch.Tensor]:
        """`load_local(start, stop)` returns this rank's [stop-start,6,H,W] device tensor."""
        ranges = [shard_range(global_batch_size, self.world, r) for r in range(self.world)]
        start, stop = ranges[self.rank]
        x = load_local(start, stop)
        if stop > start:
            flow = self.forward_fn(x)
        else:                                   # more ranks than pairs: nothing to do on this rank
            flow = x.new_zeros((0, 2, x.shape[2] // 4, x.shape[3] // 4))
        if self.world == 1:
            return flow
        return gather_flows(flow, [b - a for a, b in ranges], dst=0, group=self.group)
