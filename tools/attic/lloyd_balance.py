"""How evenly does the list-form E-step's work fall on its blocks and waves?  Needs a -DKM_BLOCKTIME=1 build:
    python tools/build_variants.py kmeans.hip:blk=-DKM_BLOCKTIME=1
    BROV2_LIBRARY=$PWD/build_variants/blk/libbrov2.so python3 tools/lloyd_balance.py
Every block adds its duration, every wave the time until it leaves its pass loop (100 MHz ticks); with one block per CU, all resident
from the start, the launch lasts as long as its slowest block: mean block time / launch time is what a perfect balance would recover."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = 10_000_000
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, 512, 500
nb = pairs // L
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
X = Xe.view(-1, n)
lib = _lib.load_library()
f = lib.brov_debug_kmblk
f.restype = ctypes.c_int
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
for rep in range(2):
    f(buf, 1)
    tm = {}
    ctx.set_timing(True)
    C, inertia, n_iter = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=300, ctx=ctx, timings=tm)
    torch.cuda.synchronize()
    ctx.set_timing(False)
    f(buf, 0)
    info = ctx.kmeans_loop_info() if hasattr(ctx, "kmeans_loop_info") else {}
    blocks, waves = max(1, buf[1]), max(1, buf[3])
    print(f"Lloyd {tm['lloyd_ms']:.1f} ms; list-form launches x blocks: {buf[1]}; mean block time {buf[0] / blocks / 100.0:.1f} us, "
          f"mean wave loop time {buf[2] / waves / 100.0:.1f} us  (compare with the launch's duration in the kernel trace) {info}", flush=True)
