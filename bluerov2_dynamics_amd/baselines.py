"""Learned double-integrator baseline of the reference's comparison scripts, on the GPU.

Reference: estimate_di_gains / simulate_double_integrator / multistep_rmse_endpoint_di in
training/train_tank_brov2_full_comparison.py:510-595 (Euler, 8 thruster inputs), ..._rk4.py:440-547 (RK4),
..._wrench_comp.py:270-361 (6-D wrench input) and ..._wrench_quat.py:301-392 (quaternion state).
The gain estimate is a 8x8 (6x6) ridge solve on the host; rollouts and the sliding-window error run in
the same kernels as the Fossen models (csrc/rollout.hip, models BROV_DI_*)."""
import numpy as np

from . import _lib, engine


def estimate_di_gains(X_train, U_train, dt, ridge=1e-3):
    """K_lin, K_ang [nu,3]: ridge least squares of forward-differenced body accelerations on the inputs
    (the velocity columns are the last six of the state, for the 12- and the 13-state layouts alike)."""
    X = np.asarray(X_train, dtype=float)
    U = np.asarray(U_train, dtype=float)
    V, W = X[:, -6:-3], X[:, -3:]
    dV = (V[1:] - V[:-1]) / max(dt, 1e-9)
    dW = (W[1:] - W[:-1]) / max(dt, 1e-9)
    G = U[:-1]
    GTG = G.T @ G
    reg = GTG + ridge * np.eye(GTG.shape[0])
    return np.linalg.solve(reg, G.T @ dV), np.linalg.solve(reg, G.T @ dW)


class DoubleIntegrator:
    """dpos = R v, dang = w (or q_dot), dv = u K_lin, dw = u K_ang."""

    def __init__(self, K_lin, K_ang, quaternion=False, device=None):
        self.K_lin = np.ascontiguousarray(K_lin, dtype=float)
        self.K_ang = np.ascontiguousarray(K_ang, dtype=float)
        nu = self.K_lin.shape[0]
        if quaternion:
            assert nu == 6, "the quaternion double integrator is wrench driven (6 inputs)"
            self.model = _lib.DI_WRENCH_QUAT
        else:
            self.model = {8: _lib.DI_THRUSTER_EULER, 6: _lib.DI_WRENCH_EULER}[nu]
        self._ctx = _lib.Context(_lib.default_context(device).device)
        self._ctx.set_di_gains(self.K_lin, self.K_ang)

    @classmethod
    def fit(cls, X_train, U_train, dt, ridge=1e-3, quaternion=False):
        return cls(*estimate_di_gains(X_train, U_train, dt, ridge), quaternion=quaternion)

    def simulate(self, x0, U_seq, dt, integrator="euler"):
        """simulate_double_integrator: (len(U_seq)+1, nx)."""
        r = engine.rollout(self.model, integrator, np.asarray(x0, float)[None], np.asarray(U_seq, float)[None], dt, ctx=self._ctx)
        return r["traj"][0]

    def rollout(self, x0, U, dt, integrator="euler", stride=1):
        return engine.rollout(self.model, integrator, x0, U, dt, stride=stride, ctx=self._ctx)

    def multistep_rmse_endpoint(self, X, U, H, dt, integrator="euler"):
        """multistep_rmse_endpoint_di."""
        return engine.window_rmse(self.model, integrator, X, U, H, dt, carry_lag=False, ctx=self._ctx)
