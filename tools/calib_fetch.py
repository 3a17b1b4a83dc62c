#!/usr/bin/env python3
"""FETCH_SIZE calibration: stream a known number of bytes (1 GiB, far beyond the 256 MiB Infinity Cache) through LDS with the
kernels' own LDS-DMA instructions.  Run under `rocprofv3 --pmc FETCH_SIZE`; tools/pmc_avg.py then gives KiB per launch of
calib_dma_read_kernel<4> (dword DMA) and <16> (16-byte DMA) to compare with 1 048 576 KiB.
usage: tools/calib_fetch.py [width ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n = 1 << 30
src = torch.rand(n // 4, device=dev)
sums = torch.empty(2048 * 256, device=dev)
widths = [int(a) for a in sys.argv[1:]] or [4, 16]
for w in widths:
    for _ in range(4):
        rc = lib.pwc_calib_lds_dma_read(src.data_ptr(), sums.data_ptr(), n, w, 2048, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    print("width %d: streamed %d KiB per launch, checksum %.3f" % (w, n // 1024, float(sums.sum())))
