#!/usr/bin/env python3
"""The reference's simulation study (training/train_sim_brov2_koopmanEDMDc.py:150-226) on the MI355X engine, in the
two shapes SURVEY.md section 8(d) names:

  single   the script as published: ONE long Euler rollout (dt = 0.05) driven by smooth random thruster commands
           u_k = clip(0.98 u_{k-1} + 0.02 xi_k), sensor noise on the logged states, 80/20 split, KoopmanEDMDc.fit,
           one-/10-/100-step RMSE, a 200-step open-loop `simulate`.  The random numbers are drawn from numpy exactly in
           the script's order (np.random.seed(42); per step 8 input normals, then 3+3+3+3 noise normals), so the data
           set is the reference's, sample for sample.
  ensemble BASELINE configs 3/4: many independent rollouts (counter-based AR(1) command stream, generated on the
           device), pairs never cross a rollout boundary (fit_multi semantics), every rank of a torch.distributed job
           rolls out and lifts its own shard, ONE all-reduce of the Gram blocks, identical host solve on every rank.

    python examples/sim_koopman.py single   [--steps 240000 --rbfs 200]
    python examples/sim_koopman.py ensemble [--rollouts 20000 --len 500 --rbfs 512]
    python -m torch.distributed.run --nproc-per-node 8 ... examples/sim_koopman.py ensemble --rollouts 1048576
"""
import argparse
import os
import sys
from time import perf_counter

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bluerov2_dynamics_amd import engine                               # noqa: E402
from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2             # noqa: E402
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc    # noqa: E402

# train_sim_brov2_koopmanEDMDc.py:174-177
NOISE_STD = np.array([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3)


def reference_dataset(N, dt=0.05, seed=42):
    """States (noisy), inputs and noiseless states of the script's data loop (:150-192), N steps.
    The rollout itself is one Euler trajectory of the thruster model on the GPU."""
    rs = np.random.RandomState(seed)                 # np.random.seed(42) + global draws == this stream
    z = rs.randn(N, 20)                              # per step: randn(8) for the input, then randn(3) x 4 for the noise
    U = np.empty((N, 8))
    u = np.zeros(8)
    for k in range(N):                               # random_input(): alpha = 0.98, noise 0.02 randn, clip
        u = np.clip(0.98 * u + 0.02 * z[k, :8], -1.0, 1.0)
        U[k] = u
    rov = BlueROV2(dt=dt)
    traj = rov.simulate(np.zeros(12), U, dt, "euler")          # [N+1, 12], x_0 = 0; the script logs x_1..x_N
    X_true = traj[1:]
    noise = np.concatenate([z[:, 8:11], z[:, 11:14], z[:, 14:17], z[:, 17:20]], axis=1)
    # the script adds: position <- z[8:11], Euler angles <- z[11:14], velocity <- z[14:17], rates <- z[17:20]
    std = np.array([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3)
    X = X_true + noise * std
    return X, U, X_true


def run_single(N=240000, dt=0.05, n_rbfs=200, gamma=1.0, ridge=1e-3, horizon=200, verbose=True):
    t = {}
    t0 = perf_counter()
    X, U, X_true = reference_dataset(N, dt)
    t["dataset"] = perf_counter() - t0
    split = int(0.8 * N)
    Xtr, Utr = X[:split], U[:split]
    Xte, Ute = X[split - 1:], U[split - 1:]          # -1 for causality (:199)
    model = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=n_rbfs, gamma=gamma, ridge=ridge)
    t0 = perf_counter()
    model.fit(Xtr, Utr)
    t["fit"] = perf_counter() - t0
    out = {"model": model, "timings": t, "split": split}
    t0 = perf_counter()
    out["rmse_1"] = model.evaluate(Xte, Ute)
    out["rmse_10"] = model.multistep_rmse(Xte, Ute, H=10)
    out["rmse_100"] = model.multistep_rmse(Xte, Ute, H=100)
    t["score"] = perf_counter() - t0
    out["pred_traj"] = model.simulate(Xte[0], Ute[:horizon])
    out["true_traj"] = X[split - 1: split - 1 + horizon + 1]
    if verbose:
        print("Model fitted!")
        print(f"One-step RMSE on test set: {out['rmse_1']:.4f}")
        print(f"10-step RMSE on test set: {out['rmse_10']:.4f}")
        print(f"100-step RMSE on test set: {out['rmse_100']:.4f}")
        print("[timing] seconds:", {k: round(v, 3) for k, v in t.items()})
    return out


def run_ensemble(rollouts=20000, L=500, dt=0.02, n_rbfs=512, gamma=1.0, ridge=1e-3, seed=0xED3D, holdout=0.1, verbose=True):
    """Configs 3/4.  Returns dict(A, B, centers, rmse_1/10/100 on this rank's held-out rollouts, timings)."""
    import torch
    import torch.distributed as tdist
    from bluerov2_dynamics_amd import _lib, dist as bdist
    rank, world = 0, 1
    if "RANK" in os.environ and not tdist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        tdist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if tdist.is_initialized():
        rank, world = tdist.get_rank(), tdist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    ctx = _lib.default_context(dev.index)
    n, r = 12, 8
    b0, b1 = bdist.shard_range(rollouts, rank, world)
    nb = b1 - b0
    t = {}
    torch.cuda.synchronize()
    t0 = perf_counter()
    U = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
    engine.fill_controls_dev(U, "btu", "ar1", seed=seed, b0=b0, T_total=L, ctx=ctx)
    X = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
    x0 = torch.zeros((nb, n), dtype=torch.float64, device=dev)
    engine.rollout_dev(_lib.THRUSTER_EULER, "euler", x0, U, dt, traj=X, layout="btu", stride=1, ctx=ctx)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    X += torch.randn(X.shape, generator=g, dtype=torch.float64, device=dev) * torch.from_numpy(NOISE_STD).to(dev)
    torch.cuda.synchronize()
    t["rollouts"] = perf_counter() - t0
    n_hold = max(1, int(holdout * nb))
    ntr = nb - n_hold
    # centres: rank 0 runs k-means over its training states, everybody receives them
    t0 = perf_counter()
    C = torch.empty((n_rbfs, n), dtype=torch.float64, device=dev)
    if rank == 0:
        Ck, _, _ = engine.kmeans_centers_dev(X[:ntr].reshape(-1, n), n_rbfs, random_state=0, ctx=ctx)
        C.copy_(Ck)
    bdist.broadcast_centers_(C)
    torch.cuda.synchronize()
    t["kmeans"] = perf_counter() - t0
    t0 = perf_counter()
    p, d = n + n_rbfs + r, n + n_rbfs
    GtG = torch.zeros((p, p), dtype=torch.float64, device=dev)
    GtY = torch.zeros((p, d), dtype=torch.float64, device=dev)
    engine.gram_dev(X[:ntr].reshape(-1, n), U[:ntr].reshape(-1, r), C, gamma, ntr, L, L + 1, L, GtG, GtY, ctx=ctx)
    bdist.allreduce_gram_(GtG, GtY)                 # the ONE collective of the fit
    torch.cuda.synchronize()
    t["gram_allreduce"] = perf_counter() - t0
    t0 = perf_counter()
    A, B = engine.solve_AB(GtG.cpu().numpy(), GtY.cpu().numpy(), ridge, d)
    t["solve"] = perf_counter() - t0
    model = KoopmanEDMDc(state_dim=n, input_dim=r, n_rbfs=n_rbfs, gamma=gamma, ridge=ridge)
    model.centers_, model.A_, model.B_, model.lift_dim_ = C.cpu().numpy(), A, B, d
    # held-out rollouts of this rank, windows inside each rollout only: score a handful as separate sequences
    t0 = perf_counter()
    Xh, Uh = X[ntr:ntr + min(n_hold, 8)].cpu().numpy(), U[ntr:ntr + min(n_hold, 8)].cpu().numpy()
    se = {1: 0.0, 10: 0.0, 100: 0.0}
    cnt = {1: 0, 10: 0, 100: 0}
    for q in range(Xh.shape[0]):
        Uq = np.vstack([Uh[q], np.zeros((1, r))])            # align U with the L+1 states; the last row is never used
        for H in (1, 10, 100):
            if L + 1 - H > 0:
                s_, _ = engine.multistep_se(Xh[q], Uq, model.centers_, gamma, A, B, H, ctx=ctx)
                se[H] += s_
                cnt[H] += (L + 1 - H) * n
    t["score"] = perf_counter() - t0
    out = {"A": A, "B": B, "centers": model.centers_, "model": model, "timings": t, "pairs_local": ntr * L, "world": world}
    for H in (1, 10, 100):
        out[f"rmse_{H}"] = float(np.sqrt(se[H] / cnt[H])) if cnt[H] else float("nan")
    if verbose and rank == 0:
        print(f"[ensemble] {world} rank(s) x {ntr} training rollouts x {L} steps = {world * ntr * L} pairs, k = {n_rbfs}")
        print(f"  RMSE on held-out rollouts: 1-step {out['rmse_1']:.5f}  10-step {out['rmse_10']:.5f}  100-step {out['rmse_100']:.5f}")
        print("  [timing] seconds:", {k: round(v, 3) for k, v in t.items()})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["single", "ensemble"])
    ap.add_argument("--steps", type=int, default=240000)
    ap.add_argument("--rollouts", type=int, default=20000)
    ap.add_argument("--len", type=int, default=500)
    ap.add_argument("--rbfs", type=int, default=None)
    ap.add_argument("--gamma", type=float, default=1.0)
    ap.add_argument("--ridge", type=float, default=1e-3)
    a = ap.parse_args()
    if a.mode == "single":
        run_single(a.steps, n_rbfs=a.rbfs or 200, gamma=a.gamma, ridge=a.ridge)
    else:
        run_ensemble(a.rollouts, a.len, n_rbfs=a.rbfs or 512, gamma=a.gamma, ridge=a.ridge)
