"""Batch-shard layer on CPU: world_size-2 gloo processes (the N>1 path of bench.py / parallel.py).
The per-rank flow function is a stand-in (the HIP forward needs a GPU); what is tested is the sharding,
the one-buffer weight broadcast and the ragged gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opticalflow_amd.parallel import shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _fake_flow(x):
    # any per-sample function: shape [b,6,H,W] -> [b,2,H/4,W/4]
    return torch.stack((x[:, :3].mean(dim=1), x[:, 3:].mean(dim=1)), 1)[:, :, ::4, ::4].contiguous()


def _worker(rank, world, port, nitems, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from opticalflow_amd.parallel import ShardedFlow, broadcast_parameters, gather_flows
        torch.manual_seed(100 + rank)                     # ranks start with DIFFERENT weights
        lin = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 3))
        nbytes = broadcast_parameters(lin, src=0)
        assert nbytes == sum(p.numel() for p in lin.parameters()) * 4
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 3))
        torch.manual_seed(100)
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 3))
        for p, q in zip(lin.parameters(), ref.parameters()):
            assert torch.equal(p, q)

        full = torch.rand(nitems, 6, 16, 32, generator=torch.Generator().manual_seed(5))
        sf = ShardedFlow(_fake_flow)
        out = sf(nitems, lambda a, b: full[a:b])
        if rank == 0:
            assert out is not None and torch.equal(out, _fake_flow(full))
        else:
            assert out is None
        # wrong bookkeeping is rejected
        try:
            gather_flows(torch.zeros(1, 2, 4, 8), [2] * world)
            raise AssertionError("count mismatch not detected")
        except ValueError:
            pass
        ok[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nitems", [8, 5, 1])
def test_sharded_flow_world2_gloo(nitems):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ok = mp.get_context("spawn").Array("i", [0, 0])
    mp.spawn(_worker, args=(2, port, nitems, ok), nprocs=2, join=True)
    assert list(ok) == [1, 1]
