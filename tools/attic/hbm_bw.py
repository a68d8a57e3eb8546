#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box with plain torch kernels (fill = write only, sum = read only, copy = both)."""
import time, torch
n = 4 * 1024**3 // 8 * 5      # 20 GiB of doubles
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
gb = n * 8 / 1e9
print("fill  (write)      %.0f GB/s" % (gb / t(lambda: x.fill_(1.0))))
print("zero  (memset)     %.0f GB/s" % (gb / t(lambda: x.zero_())))
print("sum   (read)       %.0f GB/s" % (gb / t(lambda: x.sum())))
print("copy  (read+write) %.0f GB/s moved" % (2 * gb / t(lambda: y.copy_(x))))
print("mul_  (read+write) %.0f GB/s moved" % (2 * gb / t(lambda: x.mul_(1.0000001))))
