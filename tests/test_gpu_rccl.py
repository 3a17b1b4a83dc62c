"""RCCL on the GPU box: one rank, backend "nccl" (= RCCL on ROCm).  A one-GPU box cannot hold two RCCL ranks, so this does not
measure anything over xGMI; what it pins is that every collective the N>1 paths of bench.py / parallel.py / kitti.py issue
(flat weight broadcast, gather into preallocated chunk views, barrier, float64 MAX all-reduce, the sharded KITTI stream and its
three-number score reduction) is accepted by RCCL with device tensors exactly as written -- the gloo tests cannot show that.
The rank runs in a child process so a wedged communicator cannot take the test session with it."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from opticalflow_amd import PWCDCNet, _lib
from opticalflow_amd.kitti import ShardedStream, evaluate_pairs_sharded
from opticalflow_amd.parallel import AsyncFlowGather, FlowGather, ShardedFlow, broadcast_parameters, gather_flows
from opticalflow_amd.weights import synthetic_state_dict

_lib.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

net = PWCDCNet(precision="fp32")
net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
net = net.to(dev).eval()
before = [p.detach().clone() for p in net.parameters()]
nbytes = broadcast_parameters(net, src=0)
assert nbytes == sum(p.numel() * p.element_size() for p in net.parameters()) + sum(b.numel() * b.element_size() for b in net.buffers())
assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))

x = torch.rand(3, 6, 64, 128, generator=torch.Generator().manual_seed(3)).to(dev)
flow = net(x).clone()
g = gather_flows(flow, [3], dst=0)
assert g is not None and g.is_cuda and torch.equal(g, flow)
g2 = gather_flows(flow, [3], dst=0)
assert g2.data_ptr() == g.data_ptr()                      # buffers allocated once
fg = FlowGather([2], (2, 16, 32), torch.float32, dev)
assert torch.equal(fg(flow[:2]), flow[:2])
sf = ShardedFlow(lambda t: net(t).clone())
assert torch.equal(sf(3, lambda a, b: x[a:b]), flow)

# overlapped gather (bench.py N > 1): same tensors as the synchronous form, and -- by event timestamps on this one GPU -- forward k+1
# STARTS before gather k has finished.  A one-rank gather is a copy of a few KB, so the side stream is slowed down artificially
# (a spin kernel in front of the collective) to make the order observable; the synchronous form could never show it.
net.use_graph = True
xg = net.graph_input(3, 64, 128, dev)
xg.copy_(x)
ag = AsyncFlowGather([3], (2, 16, 32), torch.float32, dev)
starts = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
flows, tickets = [], []
for k in range(4):
    starts[k].record()
    f = net(xg)
    flows.append(f.clone())
    with torch.cuda.stream(ag.side):
        torch.cuda._sleep(20_000_000)                      # ~10 ms of spinning ahead of gather k on the side stream
    tickets.append(ag.submit(f))
    if k >= 1:
        got = ag.result(tickets[k - 1])                    # waits (on the current stream) for gather k-1 only
        assert got is not None and torch.equal(got, flows[k - 1]) and torch.equal(got, flow)
ag.synchronize()
torch.cuda.synchronize()
assert torch.equal(ag.result(tickets[3]), flows[3])
lead = starts[2].elapsed_time(ag.done[1])                  # gather 1 finished this many ms AFTER forward 2 was enqueued to start
assert lead > 0.0, lead
print("overlap: gather 1 finished %%.2f ms after forward 2 started" %% lead)
net.use_graph = False

dist.barrier()
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25

rng = np.random.default_rng(0)
pairs = [(rng.integers(0, 256, (100, 150, 3), dtype=np.uint8), rng.integers(0, 256, (100, 150, 3), dtype=np.uint8)) for _ in range(3)]
stream = ShardedStream.for_model(net, 100, 150, dev, batch=2)
seen = []
for idx, full, gathered in stream.run(pairs):
    assert full.shape[0] == len(idx) and tuple(full.shape[1:]) == (2, 100, 150)
    assert gathered is not None and gathered[0] == list(idx) and gathered[1].shape[0] == len(idx)
    seen += list(idx)
assert seen == [0, 1, 2]
samples = [(a, b, np.zeros((100, 150, 2), np.float32), np.ones((100, 150), bool)) for a, b in pairs]
epe, fl, n = evaluate_pairs_sharded(stream, samples)
assert n == 3 and np.isfinite(epe) and 0.0 <= fl <= 100.0          # Fl-all is a percentage
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", nbytes)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_rccl_single_rank_collectives():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, "stdout:\n%s\nstderr:\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
