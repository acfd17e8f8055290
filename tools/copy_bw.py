#!/usr/bin/env python3
"""Measured device memory bandwidth of the box: torch copy (read + write) and a read-only reduction of 8 GiB."""
import time, torch
x = torch.empty(2 << 30, dtype=torch.float32, device='cuda').normal_()
y = torch.empty_like(x)
for name, fn, nbytes in (('copy (read+write)', lambda: y.copy_(x), 2 * x.numel() * 4), ('sum (read only)', lambda: x.sum(), x.numel() * 4)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'{name}: {nbytes / dt / 1e12:.2f} TB/s')
