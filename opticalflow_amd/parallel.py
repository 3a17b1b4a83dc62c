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


def gather_flows(local: torch.Tensor, counts: List[int], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather per-rank flow fields [b_r,2,h,w] to `dst`; returns the [sum b_r,2,h,w] batch there, None elsewhere.

    Ranks may hold different counts (ragged tail); shorter shards are padded to the longest for the
    collective and trimmed on arrival.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    if local.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d items, counts says %d" % (rank, local.shape[0], counts[rank]))
    m = max(counts)
    if m == 0:
        return local if rank == dst else None
    send = local
    if local.shape[0] < m:
        pad = local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))
        send = torch.cat((local, pad), 0)
    send = send.contiguous()
    out_device = send.device
    if send.is_cuda and dist.get_backend(group) == "gloo":
        send = send.cpu()                      # gloo has no device gather (CPU rehearsal of the RCCL path)
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=bufs, dst=dst, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(out_device)
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


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
