"""KoopmanEDMDc.fit_multi() on HOST trajectory lists at BASELINE config-3 size (20 000 bags x 501 states): the public call against the
device-resident engine.fit_dev(order="fit_multi") on the same data + the plain upload of the stacked arrays.  Run on the GPU box.

    python3 tools/time_fit_multi.py [nbags] [L] [views|copies|ragged]
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 500
mode = sys.argv[3] if len(sys.argv) > 3 else "views"
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, gamma, ridge = 12, 8, 512, 1.0, 1e-3
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
Xh = Xe.cpu().numpy()
Uh = np.zeros((nb, L + 1, r))
Uh[:, :L] = Ue.cpu().numpy()
if mode == "views":
    X_list, U_list = [Xh[b] for b in range(nb)], [Uh[b] for b in range(nb)]
elif mode == "copies":
    X_list, U_list = [Xh[b].copy() for b in range(nb)], [Uh[b].copy() for b in range(nb)]
else:       # ragged: every bag cut to a random length
    rng = np.random.default_rng(3)
    cut = rng.integers(2, L + 2, nb)
    X_list, U_list = [Xh[b, :cut[b]].copy() for b in range(nb)], [Uh[b, :cut[b]].copy() for b in range(nb)]
pairs = sum(len(x) - 1 for x in X_list)
print(f"{nb} bags, {pairs} pairs, list of {mode}", flush=True)

# plain upload of the stacked arrays (what the 1.3 x bound of the review adds to the device-resident leg)
Xs, Us = np.ascontiguousarray(Xh.reshape(-1, n)), np.ascontiguousarray(Uh.reshape(-1, r))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a_, b_ = torch.from_numpy(Xs).to(dev), torch.from_numpy(Us).to(dev)
    torch.cuda.synchronize(); up = time.perf_counter() - t0
    print(f"torch upload of the stacked arrays ({(Xs.nbytes + Us.nbytes) / 1e9:.2f} GB): {up * 1e3:.1f} ms", flush=True)
del a_, b_
for rep in range(2):
    t0 = time.perf_counter(); bt_ = engine.BagTable(X_list, U_list, n, r); print(f"BagTable (Python bookkeeping): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Xd, Ud, off = engine.upload_bags(X_list, U_list, n, r, ctx=ctx, arrays="torch")
    torch.cuda.synchronize(); ub = time.perf_counter() - t0
    print(f"engine.upload_bags: {ub * 1e3:.1f} ms", flush=True)
t0 = time.perf_counter(); vs = np.vstack(X_list); print(f"np.vstack(X_list) alone: {(time.perf_counter() - t0) * 1e3:.1f} ms"); del vs

for rep in range(3):
    tm = {}
    A, B, C = engine.fit_dev(Xd, Ud, 0, 0, k, gamma, ridge, order="fit_multi", ctx=ctx, timings=tm, bag_offsets=off)
    print(f"fit_dev(order=fit_multi, ragged, device resident): total {tm['total_s'] * 1e3:.1f} ms (centres {tm['centres_s'] * 1e3:.1f}, gram "
          f"{tm['gram_s'] * 1e3:.1f}, pinv {tm['pinv_s'] * 1e3:.1f}, P GtY {tm['apply_s'] * 1e3:.1f})", flush=True)
dev_s = tm["total_s"]
del Xd, Ud
for rep in range(3):
    m = KoopmanEDMDc(state_dim=n, input_dim=r, n_rbfs=k, gamma=gamma, ridge=ridge)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit_multi(X_list, U_list)
    host_s = time.perf_counter() - t0
    print(f"KoopmanEDMDc.fit_multi(host lists): {host_s * 1e3:.1f} ms = {pairs / host_s:.3e} samples/s; ratio to (device leg + upload) "
          f"{host_s / (dev_s + up):.2f}", flush=True)
print("A equal to the device-resident leg:", bool(np.array_equal(A, m.A_)), flush=True)
