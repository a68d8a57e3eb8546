"""KoopmanEDMDc.fit()'s device path on BASELINE config-3 data without the centres stage (centres = seeded sample rows): Gram,
host pinv, edmdc_pinv_apply_dev.  Short enough for rocprofv3 --pmc passes (tools/attic/r03_fit_pmc.sh).  Run on the GPU box.

    python3 tools/time_fit.py [pairs] [reps] [simple]
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
simple = len(sys.argv) > 3 and sys.argv[3] == "simple"
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
ctx.set_apply_variant(1 if simple else 0)
if os.environ.get("BROV2_CHUNK_ROWS"):
    ctx.check(ctx.lib.edmdc_set_chunk_rows(ctx.h, int(os.environ["BROV2_CHUNK_ROWS"])), "edmdc_set_chunk_rows")
n, r, k, gamma, ridge, L = 12, 8, 512, 1.0, 1e-3, 500
nb = max(1, pairs // L)
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * 5e-4
idx = torch.from_numpy(np.random.RandomState(0).choice(nb * (L + 1), k, replace=False)).to(dev)
C = Xe.view(-1, n)[idx].contiguous()
for rep in range(reps):
    tm = {}
    ctx.set_timing(True)
    A, B, _ = engine.fit_dev(Xe.view(-1, n), Ue.view(-1, r), nb, L, k, gamma, ridge, order="fit", centers=C, ctx=ctx, timings=tm)
    apply_ms = ctx.last_kernel_ms()
    ctx.set_timing(False)
    print(f"rep {rep}: gram {tm['gram_s'] * 1e3:.1f} ms, pinv {tm['pinv_s'] * 1e3:.1f} ms, apply {tm['apply_s'] * 1e3:.1f} ms (kernels {apply_ms:.1f}), "
          f"total {tm['total_s'] * 1e3:.1f} ms, finite {bool(np.isfinite(A).all())}", flush=True)
