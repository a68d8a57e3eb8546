"""The two caller-layout kernels of BASELINE config 4 alone, at its size (2^20 trajectories x 500 steps, [B][T][8] commands, [B][T+1][12]
states): fill_ar1_btu_kernel<8> and rollout_pair_kernel<RK4, BTU>, three launches each -- the program tools/r06_cfg4_pmc.sh puts under
rocprofv3 (every dispatch of a kernel has the same size, so per-kernel means of the counters are per-launch figures).
    python3 tools/cfg4_kernels.py [B] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bluerov2_dynamics_amd import _lib, engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
U = torch.empty((B, T, 8), dtype=torch.float64, device=dev)
X = torch.empty((B, T + 1, 12), dtype=torch.float64, device=dev)
x0 = torch.zeros((B, 12), dtype=torch.float64, device=dev)
x0[:, 2] = 5.0
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
ev[0].record()
for i in range(3):
    engine.fill_controls_dev(U, "btu", "ar1", seed=0xC0F4, b0=0, T_total=T, ctx=ctx)
    ev[1 + i].record()
for i in range(3):
    engine.rollout_dev(_lib.THRUSTER_EULER, "rk4", x0, U, 0.02, traj=X, layout="btu", stride=1, ctx=ctx)
    ev[4 + i].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]
print(f"B = {B}, T = {T}: fill_ar1_btu {min(ms[:3]):.2f} ms ({B * T * 64 / min(ms[:3]) / 1e9:.2f} TB/s written), "
      f"rollout RK4 BTU stored {min(ms[3:]):.2f} ms ({B * T * 160 / min(ms[3:]) / 1e9:.2f} TB/s algorithmic), finite: {bool(torch.isfinite(X[-1, -1]).all())}")
