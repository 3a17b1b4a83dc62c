#!/usr/bin/env python3
"""Host-side copy rates on the GPU box: what limits the staging of uint8 images into pinned memory (kitti._Ingest)."""
import time
import concurrent.futures

import numpy as np
import torch

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
H, W, B = 375, 1242, 4
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8) for _ in range(2 * B)]


def t(f, n=30):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


for label, slot in (("pageable", torch.empty((B, 2, H, W, 3), dtype=torch.uint8)), ("pinned", torch.empty((B, 2, H, W, 3), dtype=torch.uint8).pin_memory())):
    flat = slot.view(2 * B, H, W, 3)

    def serial():
        for i, im in enumerate(imgs):
            flat[i].copy_(im)
    ms = t(serial)
    assert all(torch.equal(flat[i], im) for i, im in enumerate(imgs))
    print("%-9s 8 images, torch copy_ serial:        %.3f ms (%.1f GB/s)" % (label, ms, flat.numel() / ms / 1e6))
    sn = flat.numpy()
    ins = [im.numpy() for im in imgs]

    def npserial():
        for i, im in enumerate(ins):
            np.copyto(sn[i], im)
    ms = t(npserial)
    print("%-9s 8 images, numpy copyto serial:       %.3f ms (%.1f GB/s)" % (label, ms, flat.numel() / ms / 1e6))
    for workers in (2, 4, 8):
        pool = concurrent.futures.ThreadPoolExecutor(workers)

        def par():
            list(pool.map(lambda j: np.copyto(sn[j], ins[j]), range(len(ins))))
        ms = t(par)
        print("%-9s 8 images, numpy copyto on %d threads:  %.3f ms (%.1f GB/s)" % (label, workers, ms, flat.numel() / ms / 1e6))

        def tpar():
            list(pool.map(lambda j: flat[j].copy_(imgs[j]), range(len(imgs))))
        ms = t(tpar)
        print("%-9s 8 images, torch copy_ on %d threads:   %.3f ms (%.1f GB/s)" % (label, workers, ms, flat.numel() / ms / 1e6))
        pool.shutdown()
    big = torch.cat([im.reshape(-1) for im in imgs])
    ms = t(lambda: slot.view(-1).copy_(big))
    print("%-9s one flat 11 MB torch copy_:          %.3f ms (%.1f GB/s)" % (label, ms, flat.numel() / ms / 1e6))
    if label == "pinned":
        def h2d():
            slot.to(dev, non_blocking=True)
            torch.cuda.synchronize()
        ms = t(h2d)
        print("pinned    H2D 11 MB:                            %.3f ms (%.1f GB/s)" % (ms, flat.numel() / ms / 1e6))
torch.set_num_threads(1)
slot = torch.empty((B, 2, H, W, 3), dtype=torch.uint8).pin_memory()
flat = slot.view(2 * B, H, W, 3)
ms = t(lambda: [flat[i].copy_(im) for i, im in enumerate(imgs)])
print("pinned    8 images, torch copy_ serial, 1 torch thread: %.3f ms" % ms)
