#!/usr/bin/env python3
"""Timing of the f2 path (device Lloyd iterations): 1e7 states x 12 dims, k = 512, fixed iteration count."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bluerov2_dynamics_amd import engine, _lib
from bluerov2_dynamics_amd.engine import _dptr, _hptr

N, n, k, iters = 10_000_000, 12, 512, 10
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((N, n), dtype=torch.float64, device="cuda", generator=g) * 0.5
C = X[torch.randperm(N, device="cuda", generator=g)[:k]].clone()
labels = torch.empty(N, dtype=torch.int32, device="cuda")
ctx = _lib.Context(0)
ctx.use_torch_stream()
ctx.set_timing(True)
inertia, n_iter = ctypes.c_double(0.0), ctypes.c_int(0)
torch.cuda.synchronize()
for rep in range(2):
    Cc = C.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.edmdc_kmeans_lloyd_dev(ctx.h, N, n, k, _dptr(X), n, None, _dptr(Cc), iters, 0.0, labels.data_ptr(),
                                             ctypes.byref(inertia), ctypes.byref(n_iter)), "kmeans")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("rep %d: %d iterations, wall %.1f ms, %.2f ms/iteration (+1 final assignment), inertia %.6e" % (
        rep, n_iter.value, wall * 1e3, wall * 1e3 / (n_iter.value + 1), inertia.value))

# k-means++ seeding over the same rows (scikit-learn's algorithm and random stream)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C0, idx = engine.kmeanspp_dev(X, k, mean=None, random_state=0, ctx=ctx)
    torch.cuda.synchronize()
    print("k-means++ rep %d: wall %.1f ms, device %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3, ctx.last_kernel_ms()))
