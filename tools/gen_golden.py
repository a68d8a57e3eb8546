#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: the fixtures hold inputs and the reference's outputs only.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/gen_golden.py [--only NAME]

Fixtures (all float64):
  fossen_constants.npz  Minv, allocation matrix, thruster geometry, ZOH (Ad,Bd) at 3 dt
  fossen_rhs_kat.npz    single/two-call RHS vectors for the 3 model variants (+edge cases)
  fossen_rollouts.npz   config-2 stream, first 8 trajectories, 5000 RK4 / Euler steps
                        (every 50th state) + wrench / quaternion rollouts
  windows.npz           multistep_rmse_endpoint_physics (lag carried across windows)
  edmdc.npz             KoopmanEDMDc fit / fit_multi / evaluate / multistep_rmse / simulate
  edmdc_fit.npz         KoopmanEDMDc.fit at the class defaults (k=200, ridge=1e-8) and the tank script's settings (k=500, gamma=3,
                        ridge=0.1) on 10 000 samples: the cases where fit()'s own product order matters
  edmdc_illcond.npz     KoopmanEDMDc.fit at the class defaults with a wide kernel (gamma 0.05 / 0.2) on edmdc_fit.npz's rows: cond ~ 1e14
  di.npz                learned double-integrator baseline (gains, rollouts, windowed RMSE)
  torch_rhs.npz         fossen/bluerov_torch.py bluerov_compute / ssa on random batches
  simscript.npz         training/train_sim_brov2_koopmanEDMDc.py's data loop + scores (numpy global RNG, seed 42), shortened
  cfg5w_dataset.csv.gz + cfg5w.npz  the same recording with wrench inputs through the wrench_comp / wrench_quat scripts' functions
  cfg5_dataset.csv.gz + cfg5.npz   script-level run (loader, split, Koopman / Fossen / DI RMSE table)
  cfg5_pinc.npz         the fourth row of that table: the reference's PINc evaluator with the shipped checkpoint (fixture only)
"""
import argparse
import os
import sys
import time
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BROV2_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "training"))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

from oracle.controls import controls_ar1, controls_iid  # noqa: E402

from fossen.BlueROV2 import BlueROV2 as RefThruster, ThrusterLag  # noqa: E402
from fossen.BlueROV2_thrust import BlueROV2 as RefWrenchEuler  # noqa: E402
from fossen.BlueROV2_wrench import BlueROV2 as RefWrenchQuat  # noqa: E402
import fossen.BlueROV2_wrench as refquat  # noqa: E402
from Koopman.koopmanEDMDc import KoopmanEDMDc as RefKoopman  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED_CFG2 = 0x5EED
T_CFG2 = 5000


def versions():
    import scipy
    import sklearn
    return np.array([f"numpy={np.__version__}", f"scipy={scipy.__version__}",
                     f"sklearn={sklearn.__version__}", f"python={sys.version.split()[0]}"])


# --------------------------------------------------------------------------- constants
def gen_constants():
    rov = RefThruster()
    T = np.zeros((6, 8))
    rr = np.zeros((8, 3))
    dd = np.zeros((8, 3))
    for i, th in enumerate(rov.thrusters_r):
        rr[i], dd[i] = th["r"], th["dir"]
        T[:3, i] = th["dir"]
        T[3:, i] = np.cross(th["r"], th["dir"])
    out = dict(Minv=rov.Minv, M=rov.M, W=rov.W, B=rov.B, alloc=T, thr_r=rr, thr_dir=dd,
               Ac=ThrusterLag._Ac, Bc=ThrusterLag._Bc, Cc=ThrusterLag._Cc)
    for dt in (0.01, 0.02, 0.05):
        Ad, Bd = ThrusterLag._discretise(ThrusterLag._Ac, ThrusterLag._Bc, ThrusterLag._Cc,
                                         ThrusterLag._Dc, dt)
        out[f"Ad_{dt}"] = Ad
        out[f"Bd_{dt}"] = Bd[:, 0]
    out["dts"] = np.array([0.01, 0.02, 0.05])
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "fossen_constants.npz"), **out)


# --------------------------------------------------------------------------- RHS KATs
def _rand_states(rng, n, quat=False):
    pos = rng.uniform(-3, 3, (n, 3))
    ang = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(-6, 6, n)], 1)
    nu = np.concatenate([rng.uniform(-1.5, 1.5, (n, 3)), rng.uniform(-1, 1, (n, 3))], 1)
    if not quat:
        return np.concatenate([pos, ang, nu], 1)
    q = np.stack([refquat.euler_to_quat(*a) for a in ang])
    q *= rng.uniform(0.8, 1.2, (n, 1))  # un-normalised on purpose (dynamics() normalises)
    return np.concatenate([pos, q, nu], 1)


def gen_rhs_kat():
    rng = np.random.default_rng(20251003)
    n = 48
    out = {}
    # --- thruster model: fresh object per sample, 3 consecutive calls with the same (x,u)
    X = _rand_states(rng, n)
    # edge cases: theta = +-pi/2 (cos clamp, fossen/BlueROV2.py:52-54), large yaw, zero state
    X[0, 3:6] = [0.3, np.pi / 2, 0.2]
    X[1, 3:6] = [-0.4, -np.pi / 2, 1.0]
    X[2, 3:6] = [0.1, 0.2, 300.0]
    X[3] = 0.0
    U = rng.uniform(-1, 1, (n, 8))
    U[4] = [1, -1, 1, -1, 1, -1, 1, -1]
    U[5] = 0.0
    for tag, cur, dt in (("thr", np.zeros(3), 0.02), ("thr_cur", np.array([0.3, -0.2, 0.1]), 0.05)):
        D = np.zeros((3, n, 12))
        LAG = np.zeros((3, n, 8, 3))
        TAU = np.zeros((n, 6))
        for i in range(n):
            rov = RefThruster(current_speed=cur.copy())
            for c in range(3):
                D[c, i] = rov.dynamics(X[i], U[i], dt)
                LAG[c, i] = np.stack([l._x for l in rov.thruster_lags])
            rov2 = RefThruster()
            TAU[i] = rov2.compute_thruster_forces(U[i], dt)
        out[f"{tag}_X"], out[f"{tag}_U"], out[f"{tag}_dt"] = X, U, np.float64(dt)
        out[f"{tag}_cur"], out[f"{tag}_XDOT"], out[f"{tag}_LAG"] = cur, D, LAG
        out[f"{tag}_TAU1"] = TAU
    # --- wrench, Euler angles
    Xw = X.copy()
    TAUW = rng.uniform(-1, 1, (n, 6)) * np.array([40, 40, 40, 4, 4, 4.0])
    for tag, cur in (("we", np.zeros(3)), ("we_cur", np.array([0.3, -0.2, 0.1]))):
        rov = RefWrenchEuler(current_speed=cur.copy())
        out[f"{tag}_X"], out[f"{tag}_U"], out[f"{tag}_cur"] = Xw, TAUW, cur
        out[f"{tag}_XDOT"] = np.stack([rov.dynamics(Xw[i], TAUW[i]) for i in range(n)])
    # --- wrench, quaternion
    Xq = _rand_states(rng, n, quat=True)
    Xq[0, 3:7] = 0.0  # degenerate quaternion -> identity fallback (BlueROV2_wrench.py:33-35)
    Xq[1, 3:7] = [1e-13, 0, 0, 0]
    for tag, cur in (("wq", np.zeros(3)), ("wq_cur", np.array([0.3, -0.2, 0.1]))):
        rov = RefWrenchQuat(current_speed=cur.copy())
        out[f"{tag}_X"], out[f"{tag}_U"], out[f"{tag}_cur"] = Xq, TAUW, cur
        out[f"{tag}_XDOT"] = np.stack([rov.dynamics(Xq[i], TAUW[i]) for i in range(n)])
    # --- quaternion helpers
    ang = np.stack([rng.uniform(-3, 3, 16), rng.uniform(-1.5, 1.5, 16), rng.uniform(-3, 3, 16)], 1)
    Q = np.stack([refquat.euler_to_quat(*a) for a in ang])
    out["q_euler_in"] = ang
    out["q_from_euler"] = Q
    out["q_to_euler"] = np.stack([np.array(refquat.quat_to_euler(q)) for q in Q])
    out["q_to_yaw"] = np.array([refquat.quat_to_yaw(q) for q in Q])
    out["q_to_R"] = np.stack([refquat.quat_to_rotation_matrix(q) for q in Q])
    out["q_mul"] = np.stack([refquat.quat_multiply(Q[i], Q[(i + 1) % 16]) for i in range(16)])
    out["q_deriv"] = np.stack([refquat.quat_derivative(Q[i], ang[i]) for i in range(16)])
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "fossen_rhs_kat.npz"), **out)


# --------------------------------------------------------------------------- rollouts
def _euler(rov, x0, U, dt, quat=False):
    # loop of training/train_tank_brov2_full_comparison.py:453-466 (quat: ..._wrench_quat.py:249-266)
    x = x0.copy()
    traj = [x.copy()]
    for k in range(len(U)):
        x = x + dt * rov.dynamics(x, U[k], dt)
        if quat:
            x[3:7] = refquat.quat_normalize(x[3:7])
        traj.append(x.copy())
    return np.array(traj)


def _rk4(rov, x0, U, dt, quat=False):
    # loop of training/train_tank_brov2_rk4.py:375-396.  quat=True (renormalise after the
    # full step) is OUR extension: the reference has no RK4 loop for the quaternion model.
    x = x0.copy()
    traj = [x.copy()]
    for k in range(len(U)):
        u = U[k]
        k1 = rov.dynamics(x, u, dt)
        k2 = rov.dynamics(x + 0.5 * dt * k1, u, dt)
        k3 = rov.dynamics(x + 0.5 * dt * k2, u, dt)
        k4 = rov.dynamics(x + dt * k3, u, dt)
        x = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        if quat:
            x[3:7] = refquat.quat_normalize(x[3:7])
        traj.append(x.copy())
    return np.array(traj)


def gen_rollouts():
    import train_tank_brov2_rk4 as ref_rk4
    import train_tank_brov2_full_comparison as ref_eul
    out = {}
    nb, T, dt, sub = 8, T_CFG2, 0.02, 50
    U = controls_iid(SEED_CFG2, 0, nb, T)
    x0 = np.zeros(12)
    x0[2] = 5.0
    t0 = time.time()
    R = np.zeros((nb, T // sub + 1, 12))
    E = np.zeros_like(R)
    LR = np.zeros((nb, 8, 3))
    LE = np.zeros((nb, 8, 3))
    for b in range(nb):
        rov = RefThruster(dt=dt)
        R[b] = ref_rk4.simulate_physics(x0, U[b], dt, rov)[::sub]
        LR[b] = np.stack([l._x for l in rov.thruster_lags])
        rov = RefThruster(dt=dt)
        E[b] = ref_eul.simulate_physics(x0, U[b], dt, rov)[::sub]
        LE[b] = np.stack([l._x for l in rov.thruster_lags])
        print(f"  cfg2 traj {b} done ({time.time()-t0:.0f}s)", flush=True)
    out.update(cfg2_seed=np.uint64(SEED_CFG2), cfg2_T=np.int64(T), cfg2_dt=np.float64(dt),
               cfg2_sub=np.int64(sub), cfg2_x0=x0, cfg2_rk4=R, cfg2_euler=E,
               cfg2_rk4_lag_end=LR, cfg2_euler_lag_end=LE, cfg2_U_head=U[:, :4, :])
    # AR(1) inputs (sim-script template), T=1000, dt=0.05, nonzero start, RK4 + Euler
    nb2, T2, dt2 = 4, 1000, 0.05
    U2 = controls_ar1(777, 0, nb2, T2)
    rng = np.random.default_rng(5)
    X02 = _rand_states(rng, nb2)
    X02[:, 3:5] *= 0.3
    R2 = np.zeros((nb2, T2 // 20 + 1, 12))
    E2 = np.zeros_like(R2)
    for b in range(nb2):
        R2[b] = ref_rk4.simulate_physics(X02[b], U2[b], dt2, RefThruster(dt=dt2))[::20]
        E2[b] = ref_eul.simulate_physics(X02[b], U2[b], dt2, RefThruster(dt=dt2))[::20]
    out.update(ar1_U=U2, ar1_X0=X02, ar1_dt=np.float64(dt2), ar1_sub=np.int64(20), ar1_rk4=R2, ar1_euler=E2)
    # wrench models, T=600, dt=0.02
    nb3, T3, dt3 = 4, 600, 0.02
    TAU = controls_iid(99, 0, nb3, T3, nu=6) * np.array([30, 30, 30, 3, 3, 3.0])
    Xw0 = _rand_states(rng, nb3)
    Xw0[:, 3:5] *= 0.3
    Xq0 = np.concatenate([Xw0[:, :3], np.stack([refquat.euler_to_quat(*a) for a in Xw0[:, 3:6]]), Xw0[:, 6:]], 1)
    out.update(w_TAU=TAU, w_dt=np.float64(dt3), w_sub=np.int64(20), we_X0=Xw0, wq_X0=Xq0)
    out["we_euler"] = np.stack([_euler(RefWrenchEuler(), Xw0[b], TAU[b], dt3)[::20] for b in range(nb3)])
    out["we_rk4"] = np.stack([_rk4(RefWrenchEuler(), Xw0[b], TAU[b], dt3)[::20] for b in range(nb3)])
    out["wq_euler"] = np.stack([_euler(RefWrenchQuat(), Xq0[b], TAU[b], dt3, quat=True)[::20] for b in range(nb3)])
    out["wq_rk4_ext"] = np.stack([_rk4(RefWrenchQuat(), Xq0[b], TAU[b], dt3, quat=True)[::20] for b in range(nb3)])
    # config 1: fossen/test_euler.py set-up at dt=0.02, 1000 Euler steps (SURVEY 8(d) cfg 1)
    u1 = np.array([0.1, 0.1, 0.1, 0.0, 0.5, 0.5, 0.5, 0.5])
    out["cfg1_u"] = u1
    out["cfg1_euler"] = ref_eul.simulate_physics(x0, np.tile(u1, (1000, 1)), 0.02, RefThruster())[::10]
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "fossen_rollouts.npz"), **out)


# --------------------------------------------------------------------------- windows
def _sim_dataset(N, dt, seed, noise=True):
    """Data set in the style of training/train_sim_brov2_koopmanEDMDc.py:153-197 (reference
    dynamics, AR(1) thruster commands, sensor noise)."""
    rng = np.random.default_rng(seed)
    rov = RefThruster(dt=dt)
    x = np.zeros(12)
    up = np.zeros(8)
    X = np.zeros((N, 12))
    U = np.zeros((N, 8))
    sig = np.array([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3)
    for k in range(N):
        u = np.clip(0.98 * up + 0.02 * rng.standard_normal(8), -1, 1)
        x = x + dt * rov.dynamics(x, u, dt)
        X[k] = x + (sig * rng.standard_normal(12) if noise else 0.0)
        U[k] = u
        up = u
    return X, U


def gen_windows():
    import train_tank_brov2_rk4 as ref_rk4
    import train_tank_brov2_full_comparison as ref_eul
    import train_tank_brov2_wrench_comp as ref_we
    import train_tank_brov2_wrench_quat as ref_wq
    out = {}
    N, dt = 400, 0.02
    X, U = _sim_dataset(N, dt, seed=11)
    out.update(X=X, U=U, dt=np.float64(dt), H=np.array([1, 10, 100]))
    out["thr_euler_rmse"] = np.array([ref_eul.multistep_rmse_endpoint_physics(X, U, H, dt) for H in (1, 10, 100)])
    out["thr_rk4_rmse"] = np.array([ref_rk4.multistep_rmse_endpoint_physics(X, U, H, dt) for H in (1, 10, 100)])
    # wrench data: tau = alloc @ (static thrust curve) -- any 6-D input is fine for the KAT
    rng = np.random.default_rng(3)
    TAU = np.cumsum(rng.standard_normal((N, 6)), 0) * np.array([2, 2, 2, .2, .2, .2])
    out["TAU"] = TAU
    out["we_euler_rmse"] = np.array([ref_we.multistep_rmse_endpoint_physics(X, TAU, H, dt) for H in (1, 10, 100)])
    Xq = np.concatenate([X[:, :3], np.stack([refquat.euler_to_quat(*a) for a in X[:, 3:6]]), X[:, 6:]], 1)
    out["Xq"] = Xq
    out["wq_euler_rmse"] = np.array([ref_wq.multistep_rmse_endpoint_physics(Xq, TAU, H, dt) for H in (1, 10, 100)])
    out["wq_onestep_rmse"] = np.float64(ref_wq.one_step_rmse_physics(Xq, TAU, dt))
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "windows.npz"), **out)


# --------------------------------------------------------------------------- EDMDc
def gen_edmdc():
    out = {}
    N, dt = 2000, 0.05
    X, U = _sim_dataset(N, dt, seed=42)
    k, gamma, ridge = 48, 1.0, 1e-3
    m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge)
    m.fit(X[:1600], U[:1600])
    Z = m._lift(X[:1599])
    Zp = m._lift(X[1:1600])
    G = np.hstack([Z, U[:1599]])
    out.update(X=X, U=U, n_train=np.int64(1600), k=np.int64(k), gamma=np.float64(gamma), ridge=np.float64(ridge),
               centers=m.centers_, lift64=m._lift(X[:64]), lift1=m._lift(X[7]),
               GtG=G.T @ G, GtY=G.T @ Zp, A=m.A_, B=m.B_)
    Xt, Ut = X[1600:], U[1600:]
    out["eval_rmse"] = np.float64(m.evaluate(Xt, Ut))
    out["ms_rmse"] = np.array([m.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)])
    out["sim50"] = m.simulate(Xt[0], Ut[:50])
    # fit_multi on 3 unequal bags (no cross-bag pairs, Koopman/koopmanEDMDc.py:113-152)
    cuts = [(0, 500), (500, 1300), (1300, 1600)]
    m2 = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge)
    m2.fit_multi([X[a:b] for a, b in cuts], [U[a:b] for a, b in cuts])
    out.update(multi_cuts=np.array(cuts), multi_centers=m2.centers_, multi_A=m2.A_, multi_B=m2.B_,
               multi_ms_rmse=np.array([m2.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)]))
    # gamma=3 / ridge=0.1 (config of training/train_tank_brov2_full_comparison.py:42-44)
    m3 = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=3.0, ridge=1e-1)
    m3.fit(X[:1600], U[:1600])
    out.update(g3_centers=m3.centers_, g3_A=m3.A_, g3_B=m3.B_,
               g3_ms_rmse=np.array([m3.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)]))
    out["rbf_kat"] = __import__("Koopman.koopmanEDMDc", fromlist=["_rbf_mat"])._rbf_mat(
        np.array([[0.3, -0.2, 1.0, 0.1, -0.2, 0.7, 0.4, -0.3, 0.2, 0.05, -0.1, 0.2]]),
        np.array([[0.0] * 12, [0.1] * 12]), 3.0)
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "edmdc.npz"), **out)


def gen_edmdc_fit():
    """KoopmanEDMDc.fit at the settings where the association of M = pinv(G^T G + ridge I) @ G.T @ Y matters
    (Koopman/koopmanEDMDc.py:97 evaluates it left to right): the dataclass defaults (n_rbfs=200, gamma=1, ridge=1e-8, :56-61)
    and the tank script's settings (N_RBFS=500, GAMMA=3.0, RIDGE=1e-1, training/train_tank_brov2_full_comparison.py:40-44),
    on a 10 000-sample recording-like data set (8 000 train / 2 000 test).  A and B are too large to store whole (k = 500:
    2 MB): the fixture keeps the scores, a corner block, row / column sums and norms."""
    N, ntr, dt = 10000, 8000, 0.05
    X, U = _sim_dataset(N, dt, seed=2024)
    out = dict(X=X, U=U, n_train=np.int64(ntr), dt=np.float64(dt))
    Xt, Ut = X[ntr:], U[ntr:]
    for tag, k, gamma, ridge in (("def", 200, 1.0, 1e-8), ("tank", 500, 3.0, 1e-1)):
        m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge)
        m.fit(X[:ntr], U[:ntr])
        out.update({f"{tag}_k": np.int64(k), f"{tag}_gamma": np.float64(gamma), f"{tag}_ridge": np.float64(ridge),
                    f"{tag}_centers": m.centers_,
                    f"{tag}_A_block": m.A_[:32, :32].copy(), f"{tag}_B_block": m.B_[:32].copy(),
                    f"{tag}_A_rowsum": m.A_.sum(1), f"{tag}_A_colsum": m.A_.sum(0), f"{tag}_B_colsum": m.B_.sum(0),
                    f"{tag}_A_fro": np.float64(np.linalg.norm(m.A_)), f"{tag}_B_fro": np.float64(np.linalg.norm(m.B_)),
                    f"{tag}_eval_rmse": np.float64(m.evaluate(Xt, Ut)),
                    f"{tag}_ms_rmse": np.array([m.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)]),
                    f"{tag}_train_ms_rmse": np.array([m.multistep_rmse(X[:ntr], U[:ntr], H) for H in (1, 10, 100)]),
                    f"{tag}_sim100": m.simulate(Xt[0], Ut[:100])})
        # the same data through fit_multi's association (one bag): quantifies what the order of the products is worth
        m2 = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge)
        m2.centers_ = m.centers_
        Z, Zp = m._lift(X[:ntr - 1]), m._lift(X[1:ntr])
        G = np.hstack([Z, U[:ntr - 1]])
        M2 = (np.linalg.pinv(G.T @ G + ridge * np.eye(G.shape[1])) @ (G.T @ Zp)).T
        m2.A_, m2.B_, m2.lift_dim_ = M2[:, :Z.shape[1]], M2[:, Z.shape[1]:], Z.shape[1]
        out[f"{tag}_multi_order_ms_rmse"] = np.array([m2.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)])
        out[f"{tag}_multi_order_relA"] = np.float64(np.linalg.norm(m2.A_ - m.A_) / np.linalg.norm(m.A_))
    # the ill-conditioned case: the class defaults on the 1 600-sample training set of edmdc.npz (1 599 pairs for 220
    # features, ridge 1e-8) -- here the two orders differ by 1e-6 in the H = 100 RMSE
    e = np.load(os.path.join(OUT, "edmdc.npz"))
    Xs, Us, ns = e["X"], e["U"], int(e["n_train"])
    m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=200, gamma=1.0, ridge=1e-8)
    m.fit(Xs[:ns], Us[:ns])
    Z, Zp = m._lift(Xs[:ns - 1]), m._lift(Xs[1:ns])
    G = np.hstack([Z, Us[:ns - 1]])
    M2 = (np.linalg.pinv(G.T @ G + 1e-8 * np.eye(G.shape[1])) @ (G.T @ Zp)).T
    m2 = RefKoopman(state_dim=12, input_dim=8, n_rbfs=200, gamma=1.0, ridge=1e-8)
    m2.centers_, m2.A_, m2.B_, m2.lift_dim_ = m.centers_, M2[:, :Z.shape[1]], M2[:, Z.shape[1]:], Z.shape[1]
    out.update(small_centers=m.centers_, small_A=m.A_, small_B=m.B_,
               small_ms_rmse=np.array([m.multistep_rmse(Xs[ns:], Us[ns:], H) for H in (1, 10, 100)]),
               small_multi_order_ms_rmse=np.array([m2.multistep_rmse(Xs[ns:], Us[ns:], H) for H in (1, 10, 100)]),
               small_multi_order_relA=np.float64(np.linalg.norm(m2.A_ - m.A_) / np.linalg.norm(m.A_)))
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "edmdc_fit.npz"), **out)


def gen_edmdc_illcond():
    """KoopmanEDMDc.fit at the class defaults (n_rbfs=200, ridge=1e-8, Koopman/koopmanEDMDc.py:56-61) with a WIDE kernel (gamma 0.05:
    near-duplicate RBF columns), on the training rows of edmdc_fit.npz (no new data stored): cond(G^T G + ridge I) ~ 1e14, the regime
    where the route of the p x p solve (numpy.linalg.pinv's SVD, :97, against a symmetric eigendecomposition) shows in the scores."""
    e = np.load(os.path.join(OUT, "edmdc_fit.npz"))
    X, U, ntr = e["X"], e["U"], int(e["n_train"])
    Xt, Ut = X[ntr:], U[ntr:]
    out = {}
    for tag, gamma in (("g005", 0.05), ("g02", 0.2)):
        m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=200, gamma=gamma)
        m.fit(X[:ntr], U[:ntr])
        Z = m._lift(X[:ntr - 1])
        G = np.hstack([Z, U[:ntr - 1]])
        w = np.linalg.eigvalsh(G.T @ G + m.ridge * np.eye(G.shape[1]))
        out.update({f"{tag}_gamma": np.float64(gamma), f"{tag}_ridge": np.float64(m.ridge), f"{tag}_k": np.int64(200), f"{tag}_centers": m.centers_,
                    f"{tag}_eig_min_over_max": np.float64(w.min() / w.max()), f"{tag}_eig_max": np.float64(w.max()),
                    f"{tag}_A_fro": np.float64(np.linalg.norm(m.A_)), f"{tag}_B_fro": np.float64(np.linalg.norm(m.B_)),
                    f"{tag}_train_ms_rmse": np.array([m.multistep_rmse(X[:ntr], U[:ntr], H) for H in (1, 10, 100)]),
                    f"{tag}_ms_rmse": np.array([m.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)])})
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "edmdc_illcond.npz"), **out)


# --------------------------------------------------------------------------- double integrator
def gen_di():
    """Learned double-integrator baseline of the comparison scripts: estimate_di_gains,
    simulate_double_integrator (Euler: full_comparison.py:531-573, wrench_comp.py:293-341, wrench_quat.py:324-372;
    RK4: rk4.py:497-525) and multistep_rmse_endpoint_di, on the windows.npz data set."""
    import train_tank_brov2_rk4 as ref_rk4
    import train_tank_brov2_full_comparison as ref_eul
    import train_tank_brov2_wrench_comp as ref_we
    import train_tank_brov2_wrench_quat as ref_wq
    w = np.load(os.path.join(OUT, "windows.npz"))
    X, U, TAU, Xq, dt = w["X"], w["U"], w["TAU"], w["Xq"], float(w["dt"])
    out = dict(dt=np.float64(dt), H=np.array([1, 10, 100]))
    Kl, Ka = ref_eul.estimate_di_gains(X[:300], U[:300], dt)
    out.update(thr_Klin=Kl, thr_Kang=Ka)
    out["thr_sim_euler"] = ref_eul.simulate_double_integrator(X[5], U[5:65], dt, Kl, Ka)
    out["thr_sim_rk4"] = ref_rk4.simulate_double_integrator(X[5], U[5:65], dt, Kl, Ka)
    out["thr_euler_rmse"] = np.array([ref_eul.multistep_rmse_endpoint_di(X, U, H, dt, Kl, Ka) for H in (1, 10, 100)])
    out["thr_rk4_rmse"] = np.array([ref_rk4.multistep_rmse_endpoint_di(X, U, H, dt, Kl, Ka) for H in (1, 10, 100)])
    Kl6, Ka6 = ref_we.estimate_di_gains(X[:300], TAU[:300], dt)
    out.update(we_Klin=Kl6, we_Kang=Ka6)
    out["we_sim_euler"] = ref_we.simulate_double_integrator(X[5], TAU[5:65], dt, Kl6, Ka6)
    out["we_euler_rmse"] = np.array([ref_we.multistep_rmse_endpoint_di(X, TAU, H, dt, Kl6, Ka6) for H in (1, 10, 100)])
    Klq, Kaq = ref_wq.estimate_di_gains(Xq[:300], TAU[:300], dt)
    out.update(wq_Klin=Klq, wq_Kang=Kaq)
    out["wq_sim_euler"] = ref_wq.simulate_double_integrator(Xq[5], TAU[5:65], dt, Klq, Kaq)
    out["wq_euler_rmse"] = np.array([ref_wq.multistep_rmse_endpoint_di(Xq, TAU, H, dt, Klq, Kaq) for H in (1, 10, 100)])
    out["versions"] = versions()
    np.savez(os.path.join(OUT, "di.npz"), **out)


# --------------------------------------------------------------------------- config 5 (script level)
def gen_cfg5():
    """BASELINE config 5 on synthetic data: a CSV with the reference's schema (rosbags/bag2csv.py:462-465) is run
    through the reference's own load_dataset / KoopmanEDMDc / multistep_rmse_endpoint_physics / DI helpers
    (training/train_tank_brov2_full_comparison.py:82-110, 910-912, 921-930, 982-992).  The recorded CSVs are absent
    from the reference checkout, so the data come from the reference simulator + sensor noise; a few dirty rows
    (duplicate time stamp, inf, shuffled order) exercise the loader."""
    import pandas as pd
    import train_tank_brov2_full_comparison as ref
    N, dt = 2000, 0.02
    X, U = _sim_dataset(N, dt, seed=2025)
    t = np.arange(N) * dt
    cols = ["t", "x", "y", "z", "phi", "theta", "psi", "u", "v", "w", "p", "q", "r"] + [f"u{i}" for i in range(1, 9)]
    df = pd.DataFrame(np.column_stack([t, X, U]), columns=cols)
    dirty = df.iloc[[100, 500]].copy()               # duplicated time stamps (dropped by the loader)
    dirty.iloc[0, 1] += 1.0
    bad = df.iloc[[700]].copy()
    bad["z"] = np.inf                                 # inf -> NaN -> dropped
    bad["t"] = 700.5 * dt
    df = pd.concat([df, dirty, bad]).sample(frac=1.0, random_state=1).reset_index(drop=True)   # unsorted on disk
    path = os.path.join(OUT, "cfg5_dataset.csv.gz")
    df.to_csv(path, index=False, float_format="%.12g", compression="gzip")
    Xl, Ul, dtl = ref.load_dataset(path)
    split = int(ref.TRAIN_SPLIT * len(Xl))
    Xtr, Utr, Xte, Ute = Xl[:split], Ul[:split], Xl[split:], Ul[split:]
    k = 40
    m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=ref.GAMMA, ridge=ref.RIDGE)
    m.fit(Xtr, Utr)
    Kl, Ka = ref.estimate_di_gains(Xtr, Utr, dtl, ridge=1e-3)
    table = np.array([[m.multistep_rmse(Xte, Ute, H=H) for H in (1, 10, 100)],
                      [ref.multistep_rmse_endpoint_physics(Xte, Ute, H=H, dt=dtl) for H in (1, 10, 100)],
                      [ref.multistep_rmse_endpoint_di(Xte, Ute, H=H, dt=dtl, K_lin=Kl, K_ang=Ka) for H in (1, 10, 100)]])
    np.savez(os.path.join(OUT, "cfg5.npz"), X=Xl, U=Ul, dt=np.float64(dtl), split=np.int64(split), k=np.int64(k),
             gamma=np.float64(ref.GAMMA), ridge=np.float64(ref.RIDGE), centers=m.centers_, table=table,
             rows=np.array(["Koopman", "Fossen (BlueROV2)", "Double Integrator"]), versions=versions())
    print(table)


def gen_cfg5_pinc():
    """Config 5's fourth row (fixture only; the PINc network itself is out of scope): the reference's
    multistep_rmse_endpoint_pinc (training/train_tank_brov2_full_comparison.py:866-890) with the checkpoint the checkout ships
    (models/pinc_best.pt, loaded as at :948-952) on the test split of cfg5_dataset.csv.gz, H = 1 / 10 / 100 in the script's
    order with ONE thruster-map vehicle (`rov_old`, :947: its lag state carries from one horizon's evaluation into the next,
    :992-994).  Stored next to cfg5.npz so that the ranking assertion covers four rows."""
    import torch
    import train_tank_brov2_full_comparison as ref
    g = np.load(os.path.join(OUT, "cfg5.npz"))
    X, U, dt, split = g["X"], g["U"], float(g["dt"]), int(g["split"])
    Xte, Ute = X[split:], U[split:]
    device = torch.device("cpu")
    torch.manual_seed(0)
    pinc = ref.PINcNet(hidden_sizes=ref.PINc_HIDDEN).to(device)
    pinc.load_state_dict(torch.load(os.path.join(REF, "models", "pinc_best.pt"), map_location=device))
    rov_old = RefThruster(dt=dt)
    t0 = time.time()
    row = np.array([ref.multistep_rmse_endpoint_pinc(Xte, Ute, H=H, dt=dt, model=pinc, old_model_for_map=rov_old, device=device)
                    for H in (1, 10, 100)])
    np.savez(os.path.join(OUT, "cfg5_pinc.npz"), pinc_row=row, H=np.array([1, 10, 100]), n_test=np.int64(len(Xte)),
             row_name=np.array(["PINc (ResDNN)"]), checkpoint=np.array(["models/pinc_best.pt (reference checkout), float32 forward on CPU"]),
             torch_version=np.array([torch.__version__]), versions=versions())
    print("PINc row", row, f"({time.time() - t0:.1f} s); table of the other rows:\n", g["table"])


def gen_torchrhs():
    """fossen/bluerov_torch.py: bluerov_compute and ssa on random batches (float64 and float32)."""
    import torch
    from fossen.bluerov_torch import bluerov_compute as ref_compute, ssa as ref_ssa
    rng = np.random.default_rng(9)
    x = rng.normal(size=(64, 9))
    u = rng.normal(size=(64, 4)) * 20.0
    out = dict(x=x, u=u, xdot64=ref_compute(0.0, torch.from_numpy(x), torch.from_numpy(u)).numpy(),
               xdot32=ref_compute(0.0, torch.from_numpy(x).float(), torch.from_numpy(u).float()).numpy(),
               xdot_1d=ref_compute(0.0, torch.from_numpy(x[3]), torch.from_numpy(u[3])).numpy())
    a = rng.uniform(-20, 20, 200)
    out.update(ang=a, ssa=ref_ssa(torch.from_numpy(a)).numpy(), versions=versions())
    np.savez(os.path.join(OUT, "torch_rhs.npz"), **out)


def gen_cfg5w():
    """Config 5, wrench variants: the cfg5 recording re-expressed with body-wrench inputs (columns Fx..Mz, the schema of
    rosbags/create_thrust_torque_csv.py) and run through the reference's own wrench scripts' functions:
    train_tank_brov2_wrench_comp.py (Euler-angle state) and train_tank_brov2_wrench_quat.py (quaternion state; its loader
    converts the Euler-angle file).  Wrench = allocation . thrust-curve(u), no lag -- any consistent wrench column serves
    the purpose of exercising loader + models + evaluators."""
    import pandas as pd
    import train_tank_brov2_wrench_comp as ref_we
    import train_tank_brov2_wrench_quat as ref_wq
    src = pd.read_csv(os.path.join(OUT, "cfg5_dataset.csv.gz"))
    V = src[[f"u{i}" for i in range(1, 9)]].to_numpy(float)
    F = -140.3 * V ** 9 + 389.9 * V ** 7 - 404.1 * V ** 5 + 176.0 * V ** 3 + 8.9 * V
    W = F @ np.load(os.path.join(OUT, "fossen_constants.npz"))["alloc"].T
    df = src.drop(columns=[f"u{i}" for i in range(1, 9)])
    for j, name in enumerate(["Fx", "Fy", "Fz", "Mx", "My", "Mz"]):
        df[name] = W[:, j]
    path = os.path.join(OUT, "cfg5w_dataset.csv.gz")
    df.to_csv(path, index=False, float_format="%.12g", compression="gzip")
    out = {}
    k = 40
    for tag, ref in (("we", ref_we), ("wq", ref_wq)):
        Xl, Ul, dtl = ref.load_dataset(path)
        split = int(ref.TRAIN_SPLIT * len(Xl))
        Xtr, Utr, Xte, Ute = Xl[:split], Ul[:split], Xl[split:], Ul[split:]
        m = RefKoopman(state_dim=Xl.shape[1], input_dim=6, n_rbfs=k, gamma=ref.GAMMA, ridge=ref.RIDGE)
        m.fit(Xtr, Utr)
        Kl, Ka = ref.estimate_di_gains(Xtr, Utr, dtl)
        table = np.array([[m.multistep_rmse(Xte, Ute, H=H) for H in (1, 10, 100)],
                          [ref.multistep_rmse_endpoint_physics(Xte, Ute, H=H, dt=dtl) for H in (1, 10, 100)],
                          [ref.multistep_rmse_endpoint_di(Xte, Ute, H=H, dt=dtl, K_lin=Kl, K_ang=Ka) for H in (1, 10, 100)]])
        out.update({f"{tag}_X": Xl, f"{tag}_U": Ul, f"{tag}_dt": np.float64(dtl), f"{tag}_split": np.int64(split),
                    f"{tag}_centers": m.centers_, f"{tag}_table": table, f"{tag}_gamma": np.float64(ref.GAMMA),
                    f"{tag}_ridge": np.float64(ref.RIDGE)})
        print(tag, table)
    np.savez(os.path.join(OUT, "cfg5w.npz"), k=np.int64(k), versions=versions(), **out)


def gen_simscript():
    """The data loop and scores of training/train_sim_brov2_koopmanEDMDc.py:150-226, replayed with the reference's
    classes and numpy's GLOBAL generator exactly as the script uses it (np.random.seed(42); per step randn(8) for the
    input, then randn(3) four times for the sensor noise), shortened to N steps (the script: 240 000) and k = 60 RBFs
    (the script: 200) so that the Python loop finishes in seconds.  The script itself cannot be imported: it has no
    main guard and ends in a matplotlib animation."""
    N, dt, k = 3000, 0.05, 60
    np.random.seed(42)
    rov = RefThruster(dt=dt)
    states_true, states, inputs = np.zeros((N, 12)), np.zeros((N, 12)), np.zeros((N, 8))
    x, u_prev = np.zeros(12), np.zeros(8)
    for i in range(N):
        u = np.clip(0.98 * u_prev + 0.02 * np.random.randn(rov.n_thrusters), -1.0, 1.0)
        x = x + dt * rov.dynamics(x, u, dt)
        states_true[i] = x
        ns = x.copy()
        ns[0:3] += 0.0005 * np.random.randn(3)
        ns[3:6] += 0.001 * np.random.randn(3)
        ns[6:9] += 0.0005 * np.random.randn(3)
        ns[9:12] += 0.001 * np.random.randn(3)
        states[i] = ns
        inputs[i] = u
        u_prev = u
    split = int(0.8 * N)
    Xtr, Utr = states[:split], inputs[:split]
    Xte, Ute = states[split - 1:], inputs[split - 1:]
    m = RefKoopman(state_dim=12, input_dim=8, n_rbfs=k, gamma=1.0, ridge=1e-3)
    m.fit(Xtr, Utr)
    np.savez(os.path.join(OUT, "simscript.npz"), N=np.int64(N), dt=np.float64(dt), k=np.int64(k), X=states, U=inputs, X_true=states_true,
             centers=m.centers_, A=m.A_, B=m.B_, rmse=np.array([m.evaluate(Xte, Ute), m.multistep_rmse(Xte, Ute, H=10),
                                                                 m.multistep_rmse(Xte, Ute, H=100)]),
             pred200=m.simulate(Xte[0], Ute[:200]), versions=versions())


def gen_kmeans_empty():
    """scikit-learn 1.7.2's handling of empty clusters (sklearn/cluster/_k_means_common.pyx), which the reference inherits through
    KMeans(...).fit (Koopman/koopmanEDMDc.py:85,126): expected centres / iteration counts of KMeans(init=C0, n_init=1) for
    (a) one initial centre far from all data (empty in the first iteration -> relocated to the farthest sample), (b) more clusters
    than distinct points, the reference's own call with n_init="auto", random_state=0 (exact arithmetic: all distances zero, no
    relocation), (c) `_average_centers`' in-place loop (an empty cluster in front of the biggest one takes its member SUM)."""
    import warnings
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(33)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = np.cumsum(rng.normal(0, 0.05, (3000, 12)), 0)
        C0 = X[rng.choice(len(X), 40, replace=False)].copy()
        C0[9] += 100.0
        km = KMeans(n_clusters=40, init=C0, n_init=1).fit(X)
        out.update(a_X=X, a_C0=C0, a_centers=km.cluster_centers_, a_n_iter=np.int64(km.n_iter_), a_inertia=np.float64(km.inertia_))
        P = rng.integers(-9, 10, (14, 12)).astype(float)
        m = rng.choice((2, 4, 8), len(P))
        Xd = np.concatenate([np.repeat(P, m, axis=0), np.repeat(-P, m, axis=0)])
        Xd = Xd[rng.permutation(len(Xd))]
        km = KMeans(n_clusters=40, n_init="auto", random_state=0).fit(Xd)
        out.update(b_X=Xd, b_centers=km.cluster_centers_, b_n_iter=np.int64(km.n_iter_))
        pA, pB, pC = np.array([1., 2, 0, 0]), np.array([-1., -2, 0, 0]), np.array([0., 0, 4, 0])
        Xq = np.concatenate([np.tile(pA, (2, 1)), np.tile(pB, (2, 1)), np.tile(pC, (4, 1)), np.tile(-pC, (4, 1))])
        C0q = np.array([pA, pA, pB, pC, -pC, pC])
        km = KMeans(n_clusters=6, init=C0q, n_init=1).fit(Xq)
        out.update(c_X=Xq, c_C0=C0q, c_centers=km.cluster_centers_, c_n_iter=np.int64(km.n_iter_))
    np.savez_compressed(os.path.join(OUT, "kmeans_empty.npz"), versions=versions(), **out)


GENS = dict(edmdc_illcond=gen_edmdc_illcond, kmeans_empty=gen_kmeans_empty, cfg5_pinc=gen_cfg5_pinc, torchrhs=gen_torchrhs, cfg5w=gen_cfg5w, simscript=gen_simscript, cfg5=gen_cfg5, di=gen_di, constants=gen_constants, rhs=gen_rhs_kat, rollouts=gen_rollouts, windows=gen_windows, edmdc=gen_edmdc, edmdc_fit=gen_edmdc_fit)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=list(GENS))
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for name, fn in GENS.items():
        if a.only and a.only != name:
            continue
        t0 = time.time()
        fn()
        print(f"[ok] {name} ({time.time()-t0:.1f}s)", flush=True)
