#!/usr/bin/env python3
"""Design study (not a test): how many halo pixels of a level-2 tile have taps outside the fused kernel's 23 x 52 source window, per
tile, on the up_flow the benchmark's decoder produces (CPU oracle, benchmark weights): mean 11, at most 30 of 640, 19 of 224 tiles with
none -- the numbers behind profiles/r04_corr_notes.md section 5a.   usage: python tests/flow_window_tile_hist.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flow_window_stats as S
from oracle import pwc_oracle as O
from opticalflow_amd.weights import synthetic_state_dict
torch.set_num_threads(8)
man = O.state_dict_manifest()
sd = synthetic_state_dict(man, seed=0, gain=0.85, bias_std=0.02)
x = torch.rand(2, 6, 448, 1024, generator=torch.Generator().manual_seed(1234))
with torch.no_grad():
    ups = S.up_flows(sd, x)
for lvl, flo in ups:
    if lvl != 2: continue
    B, _, H, W = flo.shape
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W) + flo[:, 0]
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1) + flo[:, 1]
    ix = ((2 * xs / max(W - 1, 1) - 1 + 1) * W - 1) / 2
    iy = ((2 * ys / max(H - 1, 1) - 1 + 1) * H - 1) / 2
    fx, fy = torch.floor(ix), torch.floor(iy)
    counts = []
    mx, my = 4, 3
    for b in range(B):
      for y0 in range(0, H, 8):
        for x0 in range(0, W, 32):
            yc, xc = min(y0 + 3, H - 1), min(x0 + 15, W - 1)
            cx, cy = fx[b, yc, xc], fy[b, yc, xc]
            ya, yb = max(y0 - 4, 0), min(y0 + 12, H)
            xa, xb = max(x0 - 4, 0), min(x0 + 36, W)
            tx, ty = fx[b, ya:yb, xa:xb], fy[b, ya:yb, xa:xb]
            out = (tx < cx - 19 - mx) | (tx + 1 > cx + 21 + mx) | (ty < cy - 7 - my) | (ty + 1 > cy + 9 + my)
            out &= ~((tx + 1 < 0) | (tx > W - 1) | (ty + 1 < 0) | (ty > H - 1))
            counts.append(int(out.sum()))
    c = torch.tensor(counts)
    print("tiles", len(c), "mean outside/tile %.1f" % c.float().mean(), "zero:", int((c == 0).sum()), "<=16:", int((c <= 16).sum()), "<=64:", int((c <= 64).sum()), ">64:", int((c > 64).sum()), "max", int(c.max()))
    print("share of outside pixels in tiles with > 64: %.3f" % (c[c > 64].sum().item() / max(c.sum().item(), 1)))
    print("pixels beyond 64 per tile (overflow), share of all outside: %.3f" % ((c - 64).clamp(min=0).sum().item() / max(c.sum().item(), 1)))
