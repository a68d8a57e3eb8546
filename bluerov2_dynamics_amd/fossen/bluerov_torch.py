"""`bluerov_compute`: the 9-state / 4-input torch right-hand side the reference's PINc physics loss calls
(fossen/bluerov_torch.py:20-67; used under torch.no_grad() in training/train_tank_brov2_full_comparison.py:747-757).

SURVEY.md section 8(a) keeps this row on PyTorch-ROCm: it is a few elementwise tensor operations inside a network's
training loop, not part of the batched rollout path, so there is no HIP kernel behind it.  It is provided so that the
scripts' `from fossen.bluerov_torch import bluerov_compute` has a counterpart in this package; it runs on whatever
device its inputs live on.

State  x = [x, y, z, cos(psi), sin(psi), u, v, w, r]   input  u = [X, Y, Z, M_z]   (body forces and yaw moment)
"""
import math

import torch

from . import parameters as P


def ssa(angle):
    """Smallest signed angle: wraps to [-pi, pi)  (fossen/bluerov_torch.py:8-18)."""
    two_pi = 2.0 * math.pi
    return angle - two_pi * torch.floor_divide(angle + math.pi, two_pi)


def bluerov_compute(t, x_, u_):
    """x_dot for a batch: x_ (B, 9) or (9,), u_ (B, 4) or (4,) -> (B, 9).  `t` is unused (ODE-solver signature).
    The sway/surge cross terms keep the signs the reference documents as "as used in the experiments"."""
    x = x_.unsqueeze(0) if x_.dim() == 1 else x_
    f = u_.unsqueeze(0) if u_.dim() == 1 else u_
    c, s = x[:, 3], x[:, 4]
    u, v, w, r = x[:, 5], x[:, 6], x[:, 7], x[:, 8]
    X, Y, Z, Mz = f[:, 0], f[:, 1], f[:, 2], f[:, 3]
    mu, mv, mw, jr = P.m - P.X_ud, P.m - P.Y_vd, P.m - P.Z_wd, P.I_zz - P.N_rd
    kin = [c * u - s * v, s * u + c * v, w, -s * r, c * r]           # d/dt of x, y, z, cos(psi), sin(psi)
    acc = [
        1 / mu * (X + mv * v * r + (P.X_u + P.X_uc * abs(u)) * u),
        1 / mv * (Y - mu * u * r + (P.Y_v + P.Y_vc * abs(v)) * v),
        1 / mw * (Z + (P.Z_w + P.Z_wc * abs(w)) * w + P.m * P.g - P.F_bouy),
        1 / jr * (Mz - (P.X_ud - P.Y_vd) * u * v + (P.N_r + P.N_rc * abs(r)) * r),
    ]
    return torch.stack(kin + acc, dim=1)
