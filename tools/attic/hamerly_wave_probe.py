"""Would whole WAVES of the sorted sample order pass a Hamerly test?  After T Lloyd iterations (our kernels) on BASELINE config-3
data: exact distances to the nearest (d1) and second nearest (d2) centre per sample; samples ordered by (label, d1) as the loop
orders them; a wave of 64 consecutive samples could skip its candidate evaluation j iterations after the bounds were refreshed
if every lane has  d1 + shift_a(j) + margin <= d2 - maxshift(j)  (upper bound grown by the own centre's movement, lower bound
shrunk by the largest movement of any centre).  Prints the fraction of such waves for j = 0, 1, 2, 5.
    python3 tools/hamerly_wave_probe.py [pairs] [T ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Ts = [int(v) for v in sys.argv[2:]] or [20, 100, 250]
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
R = float((2 * ((X - X.mean(0)) ** 2).sum(1).max()).sqrt())
margin = 1e-6 * R
for T in Ts:
    Cs = [engine.kmeans_centers_dev(X, k, random_state=0, max_iter=T + j, ctx=ctx)[0] for j in (0, 1, 2, 5)]
    C = Cs[0]
    lab = torch.empty(N, dtype=torch.int64, device=dev); d1 = torch.empty(N, dtype=torch.float64, device=dev); d2 = torch.empty(N, dtype=torch.float64, device=dev)
    c2 = (C * C).sum(1)
    for i0 in range(0, N, 1 << 20):
        xs = X[i0:i0 + (1 << 20)]
        D = ((xs * xs).sum(1)[:, None] - 2 * xs @ C.T + c2[None]).clamp_min(0)
        v, ix = torch.topk(D, 2, dim=1, largest=False)
        lab[i0:i0 + (1 << 20)] = ix[:, 0]; d1[i0:i0 + (1 << 20)] = v[:, 0].sqrt(); d2[i0:i0 + (1 << 20)] = v[:, 1].sqrt()
    order = torch.argsort(lab.double() * 1e6 + d1.clamp_max(9e5), stable=True)
    labs, d1s, d2s = lab[order], d1[order], d2[order]
    nw = N // 64
    print(f"after {T} iterations: samples with d1 + margin <= d2 (plain Hamerly test, fresh bounds): {float((d1 + margin <= d2).double().mean()):.3f}", flush=True)
    for j, Cj in zip((0, 1, 2, 5), Cs):
        sh = ((Cj - C) ** 2).sum(1).sqrt()                    # movement of every centre over j iterations
        ok = d1s + sh[labs] + margin <= d2s - sh.max()
        okw = ok[: nw * 64].view(nw, 64).all(1)
        print(f"   {j} iteration(s) after a refresh: max shift {float(sh.max()):.3e}, samples passing {float(ok.double().mean()):.3f}, "
              f"whole waves of the sorted order passing {float(okw.double().mean()):.3f}", flush=True)
