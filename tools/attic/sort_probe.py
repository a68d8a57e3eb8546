"""If the samples were sorted by (label, distance to their centre) every R Lloyd iterations, how many centres would the per-wave
candidate filter evaluate in the iterations between two sorts?  Torch simulation on config-3 data: Lloyd steps from the device
seeding; at iteration S sort; then for j = 1..R print groups per wave and the union of the candidate sets per wave (64
consecutive samples in the sorted order, labels of the previous iteration, centres of the current one).
    python3 tools/sort_probe.py [pairs] [S] [R]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, 512, 500
nb = max(1, pairs // L)
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
X = Xe.view(-1, n)
N = X.shape[0]
C, _, _ = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=S, ctx=ctx)       # centres after S iterations (our kernels)
R2 = float(2 * ((X - X.mean(0)) ** 2).sum(1).max()); margin = 1e-6 * R2 ** 0.5


def assign(C):
    lab = torch.empty(N, dtype=torch.int64, device=dev); da = torch.empty(N, dtype=torch.float64, device=dev)
    c2 = (C * C).sum(1)
    for i0 in range(0, N, 1 << 20):
        xs = X[i0:i0 + (1 << 20)]
        v, ix = ((xs * xs).sum(1)[:, None] - 2 * xs @ C.T + c2[None]).clamp_min(0).min(1)
        lab[i0:i0 + (1 << 20)] = ix; da[i0:i0 + (1 << 20)] = v.sqrt()
    return lab, da


def wave_stats(lab_prev, C):
    """groups and union of candidate sets per wave: groups by lab_prev, radius = max over the group's lanes of d(x, C[lab_prev])."""
    d_own = (X - C[lab_prev]).norm(dim=1)
    Dc = torch.cdist(C, C)
    nw = N // 64
    key = (torch.arange(nw * 64, device=dev) // 64) * k + lab_prev[: nw * 64]
    uk, inv = torch.unique(key, return_inverse=True)
    u = torch.zeros(len(uk), dtype=torch.float64, device=dev).scatter_reduce_(0, inv, d_own[: nw * 64], reduce="amax")
    a = uk % k; wave = uk // k
    groups = torch.zeros(nw, dtype=torch.float64, device=dev).index_add_(0, wave, torch.ones(len(uk), dtype=torch.float64, device=dev))
    un = torch.zeros((nw, k), dtype=torch.uint8, device=dev)
    for j0 in range(0, len(uk), 1 << 18):
        sl = slice(j0, j0 + (1 << 18))
        m = (Dc[a[sl]] < (2 * u[sl] + margin)[:, None]).to(torch.uint8)
        un.scatter_reduce_(0, wave[sl][:, None].expand(-1, k), m, reduce="amax")
    return float(groups.mean()), float(un.sum(1).double().mean()), float((groups > 8).double().mean())


lab, da = assign(C)
print(f"iteration {S}, trajectory order: groups/wave %.2f, union/wave %.1f" % wave_stats(lab, C)[:2], flush=True)
order = torch.argsort(lab.double() * 1e6 + da.clamp_max(9e5), stable=True)
X = X[order].contiguous(); lab = lab[order]
print(f"sorted by (label, radius) at iteration {S}: groups/wave %.2f, union/wave %.1f" % wave_stats(lab, C)[:2], flush=True)
for j in range(1, R + 1):
    sums = torch.zeros((k, n), dtype=torch.float64, device=dev).index_add_(0, lab, X)
    cnt = torch.zeros(k, dtype=torch.float64, device=dev).index_add_(0, lab, torch.ones(N, dtype=torch.float64, device=dev))
    C = torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1), C)
    gm, um, big = wave_stats(lab, C)                       # what the E-step of this iteration would evaluate
    nl, _ = assign(C)
    print(f"  {j:2d} iterations after the sort: groups/wave {gm:.2f}, union/wave {um:.1f}, waves with > 8 groups {big:.4f}, labels changed {int((nl != lab).sum())}", flush=True)
    lab = nl
