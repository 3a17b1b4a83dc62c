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
        # the overlapped form (bench.py's N > 1 loop): submit() per step, result(k) of the two most recent steps -- on a CPU group it
        # degenerates to the synchronous gather and must give the same tensors, step after step, ragged counts included
        from opticalflow_amd.parallel import AsyncFlowGather, shard_range
        spans = [shard_range(nitems, world, r) for r in range(world)]
        counts = [b - a for a, b in spans]
        ag = AsyncFlowGather(counts, (2, 4, 8), torch.float32, torch.device("cpu"))
        tickets = []
        for step in range(4):
            xs = torch.rand(nitems, 6, 16, 32, generator=torch.Generator().manual_seed(50 + step))
            a, b = spans[rank]
            tickets.append((ag.submit(_fake_flow(xs[a:b])), _fake_flow(xs)))
            if step >= 1:                                  # read the PREVIOUS step's result after the next submit: double buffering
                k, want = tickets[step - 1]
                got = ag.result(k)
                assert (got is None) if rank else torch.equal(got, want)
        try:
            ag.result(0)
            raise AssertionError("a result older than two submits must be refused")
        except ValueError:
            pass
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


# ---- BASELINE configs[4]: the KITTI-shaped stream sharded over ranks (kitti.ShardedStream) ---------------------
def _fake_infer(u8):
    """stand-in for GraphedInfer (the HIP forward needs a GPU): [n,2,H,W,3] uint8 -> (full [n,2,H,W], quarter [n,2,H/4,W/4])"""
    f = u8.float().mean(dim=-1)                       # [n,2,H,W]: channel 0 from image 1, channel 1 from image 2
    return f, f[:, :, ::4, ::4].contiguous()


def _stream_pairs(n):
    g = torch.Generator().manual_seed(9)
    return [(torch.randint(0, 256, (8, 16, 3), generator=g, dtype=torch.uint8),
             torch.randint(0, 256, (8, 16, 3), generator=g, dtype=torch.uint8)) for _ in range(n)]


def _stream_worker(rank, world, port, nitems, batch, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from opticalflow_amd import kitti
        pairs = _stream_pairs(nitems)
        st = kitti.ShardedStream(torch.device("cpu"), batch=batch, infer=_fake_infer, batcher=kitti.HostBatcher)
        assert list(st.my_indices(nitems)) == list(range(rank, nitems, world))
        plan = st.step_counts(nitems)
        assert sum(c[rank] for c in plan) == len(range(rank, nitems, world)) and all(max(c) <= batch for c in plan)
        seen, got = [], {}
        for idx, full, gathered in st.run(pairs):
            seen += idx
            for k, i in enumerate(idx):                                  # local full-resolution flows
                assert torch.equal(full[k], _fake_infer(torch.stack(pairs[i])[None])[0][0])
            if rank == 0:
                order, flows = gathered
                assert flows.shape[0] == len(order)
                for k, i in enumerate(order):
                    got[i] = flows[k].clone()                            # the gather buffer is reused by the next step
            else:
                assert gathered is None
        assert seen == list(range(rank, nitems, world))
        if rank == 0:                                                    # every pair of the stream arrived exactly once
            assert sorted(got) == list(range(nitems))
            for i in range(nitems):
                assert torch.equal(got[i], _fake_infer(torch.stack(pairs[i])[None])[1][0])
        # sharded evaluation (inference_kitti.py:296-314): only (sum EPE, sum Fl, n) travel
        samples = []
        for a, b in pairs:
            gt = _fake_infer(torch.stack((a, b))[None])[0][0].permute(1, 2, 0).numpy().copy()
            gt[..., 0] += 1.0                                            # EPE exactly 1 px, no outliers
            samples.append((a, b, gt, np.ones(gt.shape[:2], bool)))
        epe, fl, n = kitti.evaluate_pairs_sharded(st, samples)
        assert n == nitems and abs(epe - 1.0) < 1e-6 and fl == 0.0
        ok[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nitems,batch", [(7, 2), (4, 1), (9, 4)])
def test_kitti_sharded_stream_world2_gloo(nitems, batch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ok = mp.get_context("spawn").Array("i", [0, 0])
    mp.spawn(_stream_worker, args=(2, port, nitems, batch, ok), nprocs=2, join=True)
    assert list(ok) == [1, 1]
