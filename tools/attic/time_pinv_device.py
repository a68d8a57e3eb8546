"""pinv of the p x p regularised Gram: numpy.linalg.pinv on the host (what the reference does, Koopman/koopmanEDMDc.py:97,147) against
a symmetric eigendecomposition on the device with pinv's rcond cut-off (torch.linalg.eigh = rocSOLVER / hipSOLVER).  Timing and agreement."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import edmdc_numpy as ek

rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)
for (k, gamma, ridge, N) in ((512, 1.0, 1e-3, 40000), (500, 3.0, 0.1, 36658), (200, 1.0, 1e-8, 8000)):
    X = np.cumsum(rng.normal(0, 0.05, (N, 12)), 0)
    U = rng.uniform(-1, 1, (N, 8))
    C = X[rng.choice(N, k, replace=False)]
    G = np.hstack([ek.lift(X[:-1], C, gamma), U[:-1]])
    A = G.T @ G + ridge * np.eye(G.shape[1])
    t0 = time.perf_counter(); P = np.linalg.pinv(A); th = time.perf_counter() - t0
    Ad = torch.from_numpy(A).to(dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w, Q = torch.linalg.eigh(Ad)
        cut = 1e-15 * w.abs().max()
        winv = torch.where(w.abs() > cut, 1.0 / w, torch.zeros_like(w))
        Pd = (Q * winv) @ Q.T
        torch.cuda.synchronize(); td = time.perf_counter() - t0
    Pdh = Pd.cpu().numpy()
    Y = ek.lift(X[1:], C, gamma)
    M1 = (P @ G.T) @ Y
    M2 = (Pdh @ G.T) @ Y
    print(f"p = {A.shape[0]}, k = {k}, ridge = {ridge}: host pinv {th * 1e3:.1f} ms, device eigh-pinv {td * 1e3:.1f} ms; cond {np.linalg.cond(A):.2e}; "
          f"|P - Pd| / |P| = {np.linalg.norm(P - Pdh) / np.linalg.norm(P):.2e}; |M - Md| / |M| = {np.linalg.norm(M1 - M2) / np.linalg.norm(M1):.2e}", flush=True)
