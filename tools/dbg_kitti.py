import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from opticalflow_amd import PWCDCNet
from opticalflow_amd import kitti as K
from opticalflow_amd.weights import synthetic_state_dict
dev = torch.device("cuda:0")
net = PWCDCNet(use_graph=True).to(dev).eval()
net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
g = torch.Generator().manual_seed(0)
a = torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8)
b = torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8)
def T(name, fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print("%-28s %8.3f ms" % (name, 1e3 * (time.perf_counter() - t0) / n), flush=True); return r
host = T("stack", lambda: torch.stack((a[..., :3], b[..., :3]), 0).contiguous())
pin = torch.empty(host.shape, dtype=torch.uint8).pin_memory()
T("pin copy", lambda: pin.copy_(host))
d = T("h2d", lambda: pin.to(dev, non_blocking=True))
i1, i2 = T("normalize", lambda: K.normalize_pair(d[0], d[1]))
x, ph, pw = T("cat+pad", lambda: K.pad_to_64(torch.cat([i1, i2], 1)))
f = T("net", lambda: net(x))
f2 = T("unpad", lambda: K.unpad(f, ph, pw))
T("resize", lambda: K.flow_resize(f2, 375, 1242))
T("model_infer", lambda: K.model_infer(net, i1, i2))
pool = [(a, b)] * 50
def loop(with_model):
    t0 = time.perf_counter(); n = 0
    for p1, p2 in K.PairStream(iter(pool), dev):
        if with_model: K.model_infer(net, p1, p2)
        n += 1
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
loop(True)
print("PairStream only      %.3f ms/pair" % loop(False), flush=True)
print("PairStream + infer   %.3f ms/pair" % loop(True), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); loop(True); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
