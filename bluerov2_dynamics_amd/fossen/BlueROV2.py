"""BlueROV2 Heavy, 8-thruster model with Euler-angle state -- drop-in for the reference's
fossen/BlueROV2.py (class BlueROV2, class ThrusterLag).  The arithmetic runs in
csrc/rollout.hip; this file only keeps the object surface the training scripts use:

    rov = BlueROV2(dt=dt); xdot = rov.dynamics(x, u, dt)      (fossen/BlueROV2.py:357-400)
    tau = rov.compute_thruster_forces(u, dt)                  (:265-278)
    rov.n_thrusters, rov.thruster_lags[i]._x, rov.Minv, rov.current_speed, rov.use_tether ...

As in the reference, every dynamics()/compute_thruster_forces() call advances the eight
thruster-lag filters by one sample (the object is stateful and not re-entrant), and the
constructor's `dt` is ignored (the lag is discretised for the dt of each call).
The optional tether of the reference (off by default there) is not provided: use_tether must stay False.
"""
import numpy as np

from .. import _lib, engine
from ._vehicle import VehicleBase


def rotation_matrix(phi, theta, psi):
    """R_{b->n} = Rz(psi) Ry(theta) Rx(phi) (host helper; same convention as fossen/BlueROV2.py:23-41)."""
    cf, sf, ct, st, cp, sp = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta), np.cos(psi), np.sin(psi)
    return np.array([[cp * ct, cp * st * sf - sp * cf, sp * sf + cp * cf * st],
                     [sp * ct, cp * cf + sf * st * sp, st * sp * cf - cp * sf],
                     [-st, ct * sf, ct * cf]], dtype=float)


def euler_kinematics_matrix(phi, theta, eps=1e-7):
    """Body rates -> Euler-angle rates, with the reference's |cos(theta)| clamp (fossen/BlueROV2.py:43-62)."""
    sf, cf, st, ct = np.sin(phi), np.cos(phi), np.sin(theta), np.cos(theta)
    if abs(ct) < eps:
        ct = eps * np.sign(ct)
    t = st / ct
    return np.array([[1.0, sf * t, cf * t], [0.0, cf, -sf], [0.0, sf / ct, cf / ct]], dtype=float)


class ThrusterLag:
    """View of one row of the vehicle's [8,3] lag-state array (reference: fossen/BlueROV2.py:464-510).
    `_x` reads/writes the live state; step() is the stand-alone single-filter update."""

    _Ac = np.array([[-89.0, -72.33, -26.54], [128.0, 0.0, 0.0], [0.0, 32.0, 0.0]])
    _Bc = np.array([[8.0], [0.0], [0.0]])
    _Cc = np.array([[0.0, 5.992, 3.317]])
    _Dc = np.zeros((1, 1))

    def __init__(self, store=None, index=0):
        self._store = np.zeros((1, 3)) if store is None else store
        self._i = index
        self._dt = None
        self._Ad = self._Bd = None

    @property
    def _x(self):
        return self._store[self._i]

    @_x.setter
    def _x(self, v):
        self._store[self._i] = np.asarray(v, dtype=float).reshape(3)

    @staticmethod
    def _discretise(A, B, C, D, dt):
        p = _lib.default_params()
        for i in range(9):
            p.lag_Ac[i] = float(np.asarray(A).reshape(-1)[i])
        for i in range(3):
            p.lag_Bc[i] = float(np.asarray(B).reshape(-1)[i])
        Ad, Bd = _lib.discretise_lag(dt, p)
        return Ad, Bd.reshape(3, 1)

    def _prepare(self, dt):
        if self._dt != dt:
            self._Ad, self._Bd = self._discretise(self._Ac, self._Bc, self._Cc, self._Dc, dt)
            self._dt = dt

    def step(self, u, dt):
        self._prepare(dt)
        self._x = self._Ad @ self._x + self._Bd[:, 0] * u
        return float(self._Cc[0] @ self._x)


class BlueROV2(VehicleBase):
    MODEL = _lib.THRUSTER_EULER

    def __init__(self, rho=1000.0, current_speed=np.array([0.0, 0.0, 0.0]), dt=0.01, device=None):
        self._init_common(rho, current_speed, device)
        self.n_thrusters = 8
        p = self._params
        self.thrusters_r = [{"r": np.array(p.thr_r[i][:]), "dir": np.array(p.thr_dir[i][:])} for i in range(8)]
        self._lag = np.zeros((8, 3))
        self.thruster_lags = [ThrusterLag(self._lag, i) for i in range(8)]
        self.use_tether = False
        self.tether = None
        self.tether_state = None
        self.anchor_pos = np.zeros(3)

    def _push_extra(self, p):
        for i, th in enumerate(self.thrusters_r):
            for k in range(3):
                p.thr_r[i][k] = float(th["r"][k])
                p.thr_dir[i][k] = float(th["dir"][k])

    def _extra_key(self):
        # thruster geometry is part of the change key: editing rov.thrusters_r[i]["r" | "dir"] takes effect on the next call
        return tuple(float(v) for th in self.thrusters_r for name in ("r", "dir") for v in np.asarray(th[name], dtype=float).reshape(3))

    def _thruster_rotational_matrix(self, alpha):
        s, c = np.sin(alpha), np.cos(alpha)
        return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=float)

    def compute_thruster_forces(self, u_thrust, dt):
        """tau (6,) from normalised thruster commands; advances the lag filters (fossen/BlueROV2.py:265-278)."""
        self._sync_params()
        tau, lag = engine.thruster_forces(np.asarray(u_thrust, dtype=float).reshape(1, 8), dt, lag=self._lag[None], ctx=self._ctx)
        self._lag[...] = lag[0]
        return tau[0]

    def dynamics(self, x, u_thrust, dt):
        """xdot (12,) = f(x, u); advances the lag filters by one sample (fossen/BlueROV2.py:357-400)."""
        if self.use_tether:
            raise NotImplementedError("the tether model of the reference is not part of this engine")
        self._sync_params()
        xd, lag = engine.rhs(self.MODEL, np.asarray(x, dtype=float).reshape(1, 12), np.asarray(u_thrust, dtype=float).reshape(1, 8),
                             dt, lag=self._lag[None], ctx=self._ctx)
        self._lag[...] = lag[0]
        return xd[0]
