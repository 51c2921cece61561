#!/usr/bin/env python3
"""Host-to-device copy rate of this box from page-locked memory (what bounds the end-to-end leg once the kernels are fast):
    python profiles/h2d_rate.py"""
import time
import torch

for mb in (16, 64, 164, 512):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%4d MB page-locked -> device: %.2f ms, %.1f GB/s" % (mb, 1e3 * dt, (mb << 20) / dt / 1e9))
