"""The p x p solve of a fit at the reference's recorded shape (p = 520) and at config 3's (p = 532): numpy.linalg.pinv (SVD, what the
reference calls), a symmetric eigendecomposition on the host (numpy.linalg.eigh, pinv's cut-off), and torch.linalg.eigh on the device
at p and padded with a block c I to nearby sizes (the ROCm library's time is not monotone in p).  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import engine
from oracle import edmdc_numpy as ek

rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)


def host_eigh_pinv(A, rcond=1e-15):
    w, Q = np.linalg.eigh(0.5 * (A + A.T))
    cut = rcond * np.abs(w).max()
    winv = np.where(np.abs(w) > cut, 1.0 / np.where(w == 0, 1.0, w), 0.0)
    return (Q * winv) @ Q.T


def dev_eigh_pinv(Ad, pad_to=None, rcond=1e-15):
    p = Ad.shape[0]
    if pad_to and pad_to > p:
        c = torch.trace(Ad) / p
        B = torch.zeros((pad_to, pad_to), dtype=Ad.dtype, device=Ad.device)
        B[:p, :p] = Ad
        B[range(p, pad_to), range(p, pad_to)] = c
        Ad = B
    w, Q = torch.linalg.eigh(0.5 * (Ad + Ad.T))
    cut = rcond * w.abs().max()
    winv = torch.where(w.abs() > cut, 1.0 / w, torch.zeros_like(w))
    return ((Q * winv) @ Q.T)[:p, :p]


def best(f, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return out, min(ts[1:]) * 1e3, ts[0] * 1e3


for (k, gamma, ridge, N) in ((500, 3.0, 0.1, 36658), (512, 1.0, 1e-3, 40000), (200, 1.0, 1e-8, 8000)):
    X = np.cumsum(rng.normal(0, 0.05, (N, 12)), 0)
    U = rng.uniform(-1, 1, (N, 8))
    C = X[rng.choice(N, k, replace=False)]
    G = np.hstack([ek.lift(X[:-1], C, gamma), U[:-1]])
    A = G.T @ G + ridge * np.eye(G.shape[1])
    p = A.shape[0]
    with engine._blas_threads():
        P, th, th0 = best(lambda: np.linalg.pinv(A))
        Pe, te, te0 = best(lambda: host_eigh_pinv(A))
    Y = ek.lift(X[1:], C, gamma)
    M = (P @ G.T) @ Y
    print(f"p = {p} (k = {k}, ridge = {ridge}, cond {np.linalg.cond(A):.1e}): numpy pinv {th:.1f} ms (first {th0:.1f}); host eigh {te:.1f} ms (first {te0:.1f}), "
          f"|P - Pe| / |P| = {np.linalg.norm(P - Pe) / np.linalg.norm(P):.1e}, |M - Me| / |M| = {np.linalg.norm(M - (Pe @ G.T) @ Y) / np.linalg.norm(M):.1e}", flush=True)
    Ad = torch.from_numpy(A).to(dev)
    for pad in (None,) + tuple(q for q in (p + 1, p + 2, p + 4, (p + 7) // 8 * 8, (p + 15) // 16 * 16, (p + 31) // 32 * 32, (p + 63) // 64 * 64, 532, 544, 576, 640) if q > p):
        Pd, td, td0 = best(lambda: dev_eigh_pinv(Ad, pad))
        Pdh = Pd.cpu().numpy()
        print(f"    device eigh at {pad or p}: {td:.1f} ms (first {td0:.1f}); |P - Pd| / |P| = {np.linalg.norm(P - Pdh) / np.linalg.norm(P):.1e}, "
              f"|M - Md| / |M| = {np.linalg.norm(M - (Pdh @ G.T) @ Y) / np.linalg.norm(M):.1e}", flush=True)
