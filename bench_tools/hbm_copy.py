#!/usr/bin/env python
"""Developer tool: achievable HBM rates on this device for buffer sizes of the generator's activations
(read + write of equal size = a copy; write only = a fill; read only = a sum), torch kernels, HIP events."""
import torch
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (67, 134, 268, 1024):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x)); tf = t(lambda: y.fill_(1.0)); ts = t(lambda: x.sum())
    print(f'{mb:5d} MB: copy {2 * mb / tc / 1e6:5.2f} TB/s ({tc * 1e6:6.1f} us)  fill {mb / tf / 1e6:5.2f} TB/s  read(sum) {mb / ts / 1e6:5.2f} TB/s')
