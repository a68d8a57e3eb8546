#!/usr/bin/env python3
"""Kernel times of one EDMDc fit (10^7 pairs, k = 512) split into lift and Gram (HIP events around gram_dev; rocprof for the split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bluerov2_dynamics_amd import engine, _lib
n, r, k, nb, L = 12, 8, 512, 20000, 500
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn((nb, L + 1, n), generator=g, dtype=torch.float64, device=dev) * 0.3
U = torch.rand((nb, L, r), generator=g, dtype=torch.float64, device=dev) * 2 - 1
C = torch.randn((k, n), generator=g, dtype=torch.float64, device=dev) * 0.3
p, d = n + k + r, n + k
G = torch.zeros((p, p), dtype=torch.float64, device=dev); Y = torch.zeros((p, d), dtype=torch.float64, device=dev)
ctx = _lib.default_context(0)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    engine.gram_dev(X.view(-1, n), U.view(-1, r), C, 1.0, nb, L, L + 1, L, G, Y, ctx=ctx)
    e1.record(); torch.cuda.synchronize()
    print("%s rep %d: lift+gram %.2f ms" % (os.path.basename(os.path.dirname(os.environ.get("BROV2_LIBRARY", "shipped/libbrov2.so"))), rep, e0.elapsed_time(e1)))
