"""How many centres would a neighbour-list E-step evaluate per wave on BASELINE config-3 data?  After `iters` plain Lloyd
iterations (our kernels): per wave of 64 consecutive samples, group the lanes by label; a group with label a and radius
u = max d(x, c_a) must look at the centres with d(c_a, c) < 2 u + margin.  Prints groups per wave and candidates per wave
(the plain E-step evaluates k = 512 per wave).   python3 tools/nbr_probe.py [pairs] [iters ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
its = [int(v) for v in sys.argv[2:]] or [1, 20, 100]
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
for it in its:
    C, _, _ = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=it, ctx=ctx)
    # labels for these centres (chunked, torch)
    lab = torch.empty(N, dtype=torch.int64, device=dev); da = torch.empty(N, dtype=torch.float64, device=dev)
    c2 = (C * C).sum(1)
    for i0 in range(0, N, 1 << 20):
        xs = X[i0:i0 + (1 << 20)]
        D = ((xs * xs).sum(1)[:, None] - 2 * xs @ C.T + c2[None]).clamp_min(0)
        v, ix = D.min(1)
        lab[i0:i0 + (1 << 20)] = ix; da[i0:i0 + (1 << 20)] = v.sqrt()
    if os.environ.get("SORT_BY_LABEL") == "1":          # what a label-sorted copy of the samples would give
        order = torch.argsort(lab, stable=True)
        lab, da = lab[order], da[order]
    if os.environ.get("SORT_BY_LABEL") == "2":          # sorted by (label, distance to the centre): waves of homogeneous radius
        order = torch.argsort(lab.double() * 1e6 + da.clamp_max(9e5), stable=True)
        lab, da = lab[order], da[order]
    # per-sample need (no wave effects): centres within 2 d_i + margin of the sample's own centre
    Dc_ = torch.cdist(C, C); Ds_, _ = Dc_.sort(1)
    need = torch.zeros(N, dtype=torch.float64, device=dev)
    for i0 in range(0, N, 1 << 20):
        sl = slice(i0, i0 + (1 << 20))
        need[sl] = (Ds_[lab[sl]] < (2 * da[sl] + margin)[:, None]).sum(1).double()
    print(f"                      per-sample candidates mean {need.mean():.1f} (median/p90/p99 {torch.quantile(need[::16], torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev)).tolist()})", flush=True)
    if os.environ.get("NORM") == "1":
        # would a norm annulus prune further?  c can only win if | |x| - |c| | < d(x, c_a)  (centred coordinates)
        mu = X.mean(0)
        xn_all = torch.empty(N, dtype=torch.float64, device=dev)
        for i0 in range(0, N, 1 << 20):
            xn_all[i0:i0 + (1 << 20)] = ((X[i0:i0 + (1 << 20)] - mu) ** 2).sum(1).sqrt()
        xn_ = xn_all[order] if os.environ.get("SORT_BY_LABEL") in ("1", "2") else xn_all
        cnorm = ((C - mu) ** 2).sum(1).sqrt()
        need2 = torch.zeros(N, dtype=torch.float64, device=dev)
        for i0 in range(0, N, 1 << 19):
            sl = slice(i0, i0 + (1 << 19))
            tri = Dc_[lab[sl]] < (2 * da[sl] + margin)[:, None]
            ann = (cnorm[None, :] - xn_[sl][:, None]).abs() < (da[sl] + margin)[:, None]
            need2[sl] = (tri & ann).sum(1).double()
        print(f"                      per-sample candidates with the norm annulus: mean {need2.mean():.1f}", flush=True)
        nwv = N // 64
        labw = lab[: nwv * 64].view(nwv, 64); daw = da[: nwv * 64].view(nwv, 64); xnw = xn_[: nwv * 64].view(nwv, 64)
        uni = (labw == labw[:, :1]).all(1)                     # one-label waves
        uu = daw.max(1).values; lo = xnw.min(1).values - uu - margin; hi = xnw.max(1).values + uu + margin
        tot_t, tot_b = 0.0, 0.0
        for w0 in range(0, nwv, 1 << 15):
            sl = slice(w0, w0 + (1 << 15))
            tri = Dc_[labw[sl, 0]] < (2 * uu[sl] + margin)[:, None]
            ann = (cnorm[None, :] > lo[sl][:, None]) & (cnorm[None, :] < hi[sl][:, None])
            m = uni[sl]
            tot_t += float(tri[m].sum()); tot_b += float((tri & ann)[m].sum())
        print(f"                      one-label waves ({float(uni.double().mean()):.3f} of all): candidates/wave {tot_t / float(uni.sum()):.1f} -> "
              f"{tot_b / float(uni.sum()):.1f} with the annulus", flush=True)
    Dc = torch.cdist(C, C)
    Ds, _ = Dc.sort(1)
    nw = N // 64
    key = (torch.arange(nw * 64, device=dev) // 64) * k + lab[: nw * 64]
    uk, inv = torch.unique(key, return_inverse=True)
    u = torch.zeros(len(uk), dtype=torch.float64, device=dev).scatter_reduce_(0, inv, da[: nw * 64], reduce="amax")
    a = uk % k
    cnt = (Ds[a] < (2 * u + margin)[:, None]).sum(1)
    wave = uk // k
    groups = torch.zeros(nw, dtype=torch.float64, device=dev).index_add_(0, wave, torch.ones(len(uk), dtype=torch.float64, device=dev))
    cands = torch.zeros(nw, dtype=torch.float64, device=dev).index_add_(0, wave, cnt.double())
    # union over the groups of a wave (what a wave-uniform scan over the union of the candidate sets would evaluate)
    un = torch.zeros((nw, k), dtype=torch.uint8, device=dev)
    for j0 in range(0, len(uk), 1 << 18):
        sl = slice(j0, j0 + (1 << 18))
        m = (Dc[a[sl]] < (2 * u[sl] + margin)[:, None]).to(torch.uint8)
        un.scatter_reduce_(0, wave[sl][:, None].expand(-1, k), m, reduce="amax")
    usz = un.sum(1).double()
    q = torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev)
    print(f"                      union/wave mean {usz.mean():.1f} (median/p90/p99 {torch.quantile(usz, q).tolist()})", flush=True)
    print(f"after {it:3d} iterations: groups/wave mean {groups.mean():.2f} (median/p90/p99 {torch.quantile(groups, q).tolist()}), "
          f"candidates/wave mean {cands.mean():.1f} (median/p90/p99 {torch.quantile(cands, q).tolist()}), per group mean {cnt.double().mean():.1f}; "
          f"waves with > 8 groups {float((groups > 8).double().mean()):.4f}", flush=True)
