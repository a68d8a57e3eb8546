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


class _ThrusterEntry(dict):
    """thrusters_r[i] of the reference is a dict {"r": array(3), "dir": array(3)} (fossen/BlueROV2.py:172-232).  Here the two
    arrays are views of one row of the vehicle's [8,2,3] geometry block, and assigning a new array to either key copies it
    into that view: edits in place and by assignment both land in the block, whose bytes the per-call check compares."""

    def __init__(self, block_row):
        dict.__init__(self, r=block_row[0], dir=block_row[1])
        self._row = block_row

    def __setitem__(self, key, value):
        if key == "r" or key == "dir":
            self._row[0 if key == "r" else 1][...] = np.asarray(value, dtype=float).reshape(3)
        else:
            dict.__setitem__(self, key, value)

    # dict.update / setdefault / |= bypass __setitem__ in CPython: route them through it, so that no foreign array can take
    # the place of a view of the geometry block (the edit would never reach the device and the per-call check would see no change)
    def update(self, *args, **kw):
        for k, v in dict(*args, **kw).items():
            self[k] = v

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return self[key]

    def __ior__(self, other):
        self.update(other)
        return self

    def _keep(self, key):
        if key in ("r", "dir"):
            raise KeyError(f"thruster entry: {key!r} is a view of the vehicle's geometry block and cannot be removed")

    def __delitem__(self, key):
        self._keep(key)
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        self._keep(key)
        return dict.pop(self, key, *default)

    def popitem(self):
        raise KeyError("thruster entry: items cannot be removed")

    def clear(self):
        raise KeyError("thruster entry: items cannot be removed")


class _ThrusterList(list):
    def __setitem__(self, i, entry):
        if isinstance(i, slice):
            raise TypeError("thrusters_r: assign one thruster at a time")
        self[i]["r"] = entry["r"]
        self[i]["dir"] = entry["dir"]


class BlueROV2(VehicleBase):
    MODEL = _lib.THRUSTER_EULER

    def __init__(self, rho=1000.0, current_speed=np.array([0.0, 0.0, 0.0]), dt=0.01, device=None):
        self._init_common(rho, current_speed, device)
        self.n_thrusters = 8
        p = self._params
        self._geom = np.array([[p.thr_r[i][:], p.thr_dir[i][:]] for i in range(8)], dtype=float)      # [8][r | dir][3]
        self._geom_pushed = None
        self._thrusters = _ThrusterList(_ThrusterEntry(self._geom[i]) for i in range(8))
        self._lag = np.zeros((8, 3))
        self.thruster_lags = [ThrusterLag(self._lag, i) for i in range(8)]
        self.use_tether = False
        self.tether = None
        self.tether_state = None
        self.anchor_pos = np.zeros(3)

    @property
    def thrusters_r(self):
        return self._thrusters

    @thrusters_r.setter
    def thrusters_r(self, entries):
        entries = list(entries)
        if len(entries) != 8:
            raise ValueError("thrusters_r holds the 8 thrusters of the BlueROV2 Heavy")
        for i, e in enumerate(entries):
            self._thrusters[i] = e

    def _push_extra(self, p):
        for i in range(8):
            for k in range(3):
                p.thr_r[i][k] = float(self._geom[i, 0, k])
                p.thr_dir[i][k] = float(self._geom[i, 1, k])

    def _extra_key(self):
        # thruster geometry is part of the change key: editing rov.thrusters_r[i]["r" | "dir"] takes effect on the next call
        return self._geom.tobytes()

    def _extra_clean(self):
        return self._geom.tobytes() == self._geom_pushed

    def _extra_mark_clean(self):
        self._geom_pushed = self._geom.tobytes()

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
        return self._rhs_single(x, u_thrust, dt, lag=self._lag)      # the lag state [8,3] is advanced in place
