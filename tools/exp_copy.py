#!/usr/bin/env python3
"""Scratch: device copy / read ceilings on this box (torch kernels), for comparison with the stencil kernel."""
import torch
n = 512 ** 3
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: y.copy_(x)); print("copy 537MB->537MB: %.1f us  %.0f GB/s (r+w)" % (us, 2 * n * 4 / us / 1e3))
us = t(lambda: torch.add(x, 1.0, out=y)); print("add  : %.1f us  %.0f GB/s (r+w)" % (us, 2 * n * 4 / us / 1e3))
us = t(lambda: x.sum()); print("sum (read only): %.1f us  %.0f GB/s" % (us, n * 4 / us / 1e3))
us = t(lambda: y.fill_(1.0)); print("fill (write only): %.1f us  %.0f GB/s" % (us, n * 4 / us / 1e3))
