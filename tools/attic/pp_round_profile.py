"""k-means++ seeding once on config-3 data (or random data: argv[1] = random); run under rocprofv3 --kernel-trace to get the duration of
every pp_round launch by round (tools/pp_round_times.py reads the trace)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, 512, 500
nb = 20000
if len(sys.argv) > 1 and sys.argv[1] == "random":
    X = torch.from_numpy(np.random.default_rng(0).normal(0, 1, (nb * (L + 1), n))).to(dev)
else:
    Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
    engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
    Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
    engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
    Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
    X = Xe.view(-1, n)
ctx.set_timing(True)
for rep in range(2):
    C, idx = engine.kmeanspp_dev(X, k, mean=X.mean(0).cpu().numpy(), random_state=0, ctx=ctx)
    print("kmeanspp ms", ctx.last_kernel_ms(), flush=True)
