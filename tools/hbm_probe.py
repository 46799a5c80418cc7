#!/usr/bin/env python
"""HBM probe on the GPU box: pure write (fill), pure read (sum), copy, and two interleaved write streams of the sizes the gas
optics kernels produce -- the write-side ceiling the store-heavy kernels should be compared with."""
import torch
dev = "cuda:0"
n = 16384*140*256                      # one cell array of C4 (4.7 GB in fp64)
a = torch.empty(n, dtype=torch.float64, device=dev); b = torch.empty_like(a)
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
GB = n*8/1e9
t = timed(lambda: a.fill_(1.0)); print(f"fill   {GB:.2f} GB written           {t:.3f} ms  {GB/t:.2f} TB/s")
t = timed(lambda: (a.fill_(1.0), b.fill_(2.0))); print(f"fill x2 {2*GB:.2f} GB written          {t:.3f} ms  {2*GB/t:.2f} TB/s")
t = timed(lambda: a.sum()); print(f"sum    {GB:.2f} GB read              {t:.3f} ms  {GB/t:.2f} TB/s")
t = timed(lambda: b.copy_(a)); print(f"copy   {GB:.2f} GB read + {GB:.2f} written {t:.3f} ms  {2*GB/t:.2f} TB/s")
t = timed(lambda: torch.add(a, 1.0, out=b)); print(f"add    {GB:.2f} GB read + {GB:.2f} written {t:.3f} ms  {2*GB/t:.2f} TB/s")
