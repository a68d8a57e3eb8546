"""ctypes front-end of oracle/brov2_oracle.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

MODEL_THRUSTER_EULER, MODEL_WRENCH_EULER, MODEL_WRENCH_QUAT = 0, 1, 2
MODEL_DI_THRUSTER_EULER, MODEL_DI_WRENCH_EULER, MODEL_DI_WRENCH_QUAT = 3, 4, 5
INTEG_EULER, INTEG_RK4 = 0, 1
LAG_PER_CALL, LAG_PER_STEP = 0, 1
NX = {0: 12, 1: 12, 2: 13, 3: 12, 4: 12, 5: 13}
NU = {0: 8, 1: 6, 2: 6, 3: 8, 4: 6, 5: 6}

_dp = ctypes.POINTER(ctypes.c_double)
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "brov2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_rollout.restype = None
        _lib.orc_rhs.restype = None
        _lib.orc_window_endpoint_se.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _cur(cur):
    return None if cur is None else _c(cur, (3,))


def set_di_gains(K_lin, K_ang):
    """Gains of the double-integrator models 3..5 (global in the oracle library): K_lin, K_ang [nu,3]."""
    K_lin, K_ang = _c(K_lin), _c(K_ang)
    lib().orc_set_di_gains(int(K_lin.shape[0]), _p(K_lin), _p(K_ang))


def estimate_di_gains(X, U, dt, ridge=1e-3):
    """estimate_di_gains (training/train_tank_brov2_full_comparison.py:510-528; velocity columns are the last six)."""
    X, U = np.asarray(X, float), np.asarray(U, float)
    V, W = X[:, -6:-3], X[:, -3:]
    dV = (V[1:] - V[:-1]) / max(dt, 1e-9)
    dW = (W[1:] - W[:-1]) / max(dt, 1e-9)
    G = U[:-1]
    GTG = G.T @ G
    I = np.eye(GTG.shape[0])
    return np.linalg.solve(GTG + ridge * I, G.T @ dV), np.linalg.solve(GTG + ridge * I, G.T @ dW)


def constants():
    Minv, alloc, r, d = np.zeros(6), np.zeros((6, 8)), np.zeros((8, 3)), np.zeros((8, 3))
    lib().orc_constants(_p(Minv), _p(alloc), _p(r), _p(d))
    return dict(Minv=Minv, alloc=alloc, thr_r=r, thr_dir=d)


def discretise_lag(dt):
    Ad, Bd = np.zeros((3, 3)), np.zeros(3)
    lib().orc_discretise_lag(ctypes.c_double(dt), _p(Ad), _p(Bd))
    return Ad, Bd


def rhs(model, x, u, dt=0.02, lag=None, current=None):
    """Batched dynamics(); returns (xdot, lag_after)."""
    x = _c(x).reshape(-1, NX[model])
    u = _c(u).reshape(-1, NU[model])
    B = x.shape[0]
    lag = np.zeros((B, 8, 3)) if lag is None else _c(lag).reshape(B, 8, 3).copy()
    out = np.zeros_like(x)
    cur = _cur(current)
    lib().orc_rhs(model, _p(cur), ctypes.c_long(B), _p(x), _p(u), ctypes.c_double(dt), _p(lag), _p(out))
    return out, lag


def thruster_forces(u, dt=0.02, lag=None):
    u = _c(u).reshape(-1, 8)
    B = u.shape[0]
    lag = np.zeros((B, 8, 3)) if lag is None else _c(lag).reshape(B, 8, 3).copy()
    tau = np.zeros((B, 6))
    lib().orc_thruster_forces(ctypes.c_long(B), _p(u), ctypes.c_double(dt), _p(lag), _p(tau))
    return tau, lag


def rollout(model, integ, x0, U, dt, lag=None, lag_mode=LAG_PER_CALL, sub=1, current=None,
            store=True, nthreads=1):
    """x0 [B,nx], U [B,T,nu] -> dict(traj [B,T/sub+1,nx] | None, xT [B,nx], lag [B,8,3])."""
    U = _c(U)
    B, T, nu = U.shape
    assert nu == NU[model]
    x0 = _c(x0).reshape(B, NX[model])
    lag = np.zeros((B, 8, 3)) if lag is None else _c(lag).reshape(B, 8, 3).copy()
    traj = np.zeros((B, T // sub + 1, NX[model])) if store else None
    xT = np.zeros((B, NX[model]))
    cur = _cur(current)
    lib().orc_rollout(model, integ, lag_mode, _p(cur), ctypes.c_long(B), ctypes.c_long(T), ctypes.c_double(dt),
                      _p(x0), _p(U), _p(lag), _p(traj), ctypes.c_long(sub), _p(xT), int(nthreads))
    return dict(traj=traj, xT=xT, lag=lag)


def window_endpoint_se(model, integ, X, U, H, dt, carry_lag=True, current=None):
    """Returns (se_total, per_window[N-H]); rmse = sqrt(se_total / ((N-H)*nx))."""
    X = _c(X).reshape(-1, NX[model])
    U = _c(U).reshape(-1, NU[model])
    N = X.shape[0]
    per = np.zeros(max(N - H, 0))
    se = ctypes.c_double(0.0)
    cur = _cur(current)
    lib().orc_window_endpoint_se(model, integ, _p(cur), ctypes.c_long(N), ctypes.c_long(H), ctypes.c_double(dt),
                                 _p(X), _p(U), int(bool(carry_lag)), ctypes.byref(se), _p(per))
    return se.value, per


def window_rmse(model, integ, X, U, H, dt, carry_lag=True, current=None):
    se, per = window_endpoint_se(model, integ, X, U, H, dt, carry_lag, current)
    n = len(per)
    return float(np.sqrt(se / (n * NX[model]))) if n > 0 else float("nan")
