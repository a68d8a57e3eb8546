"""Shared host-side plumbing of the three drop-in vehicle classes: parameter struct <-> attributes,
one brov_ctx per object, batched helpers."""
import os

import numpy as np

from .. import _lib, engine

_ATTR6 = ("Xu_dot", "Yv_dot", "Zw_dot", "Kp_dot", "Mq_dot", "Nr_dot")
_LIN6 = ("Xu", "Yv", "Zw", "Kp", "Mq", "Nr")
# attributes whose assignment has to reach the device before the next call
_TRACKED = frozenset(("rho", "m", "g", "volume", "xb", "yb", "zb", "Ix", "Iy", "Iz", "current_speed") + _ATTR6 + _LIN6 +
                     tuple(l + "_abs" for l in _LIN6))


class VehicleBase:
    """Holds the vehicle constants as plain attributes (same names as the reference objects,
    fossen/BlueROV2.py:81-140) and mirrors them into the device context before each call."""
    MODEL = None
    _dirty = True
    _cs_pushed = None

    def __setattr__(self, name, value):
        # the reference reads its attributes on every dynamics() call, so an assignment takes effect on the next one; here it
        # marks the object dirty, and a call on a clean object skips the comparison of all constants (the per-call entry
        # points are latency: 40 attribute reads cost as much as the launch)
        if name in _TRACKED:
            object.__setattr__(self, "_dirty", True)
        object.__setattr__(self, name, value)

    def _init_common(self, rho, current_speed, device=None):
        if device is None:
            device = int(os.environ.get("BROV2_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._ctx = _lib.Context(device)
        p = self._ctx.get_params()
        self.rho = rho
        self.g, self.m, self.volume = p.g, p.m, p.volume
        self.xg = self.yg = self.zg = 0.0
        self.xb, self.yb, self.zb = p.xb, p.yb, p.zb
        self.Ix, self.Iy, self.Iz = p.Ix, p.Iy, p.Iz
        for i, (a, l) in enumerate(zip(_ATTR6, _LIN6)):
            setattr(self, a, p.added_mass[i])
            setattr(self, l, p.lin_damp[i])
            setattr(self, l + "_abs", p.quad_damp[i])
        self.current_speed = current_speed
        self._pushed = None
        self._params = p

    # Derived quantities.  The reference stores W, B, MRB, MA, M, Minv as plain attributes computed once in __init__ and reads
    # them on every dynamics() call (fossen/BlueROV2.py:96-126,340-355,391).  Here the device holds the primary constants
    # (m, g, rho, volume, inertias, added mass), so the derived ones are views of those: reading always reflects the
    # current primaries; B can be assigned (it maps onto `volume`); the others refuse assignment instead of silently
    # ignoring it.
    @property
    def W(self):
        return self.m * self.g

    @W.setter
    def W(self, v):
        raise AttributeError("W is derived (m * g): set m or g")

    @property
    def B(self):
        return self.rho * self.g * self.volume

    @B.setter
    def B(self, v):
        self.volume = float(v) / (self.rho * self.g)

    @property
    def MRB(self):
        return np.diag([self.m, self.m, self.m, self.Ix, self.Iy, self.Iz]).astype(float)

    @property
    def MA(self):
        return np.diag([-getattr(self, a) for a in _ATTR6]).astype(float)

    @property
    def M(self):
        return self.MRB + self.MA

    @property
    def Minv(self):
        return np.linalg.inv(self.M)

    def _derived_readonly(self, name):
        raise AttributeError(f"{name} is derived from m, Ix..Iz and the added-mass attributes: set those")

    MRB = MRB.setter(lambda self, v: self._derived_readonly("MRB"))
    MA = MA.setter(lambda self, v: self._derived_readonly("MA"))
    M = M.setter(lambda self, v: self._derived_readonly("M"))
    Minv = Minv.setter(lambda self, v: self._derived_readonly("Minv"))

    def _sync_params(self):
        """Push attribute values that differ from what the device context holds.  Fast path: nothing was assigned since the
        last push and the arrays that can be edited in place (current_speed, the thruster geometry) still hold the same bytes."""
        cs = self.current_speed
        csb = b"" if cs is None else np.asarray(cs, dtype=float).tobytes()
        if not self._dirty and csb == self._cs_pushed and self._extra_clean():
            return
        cur = np.zeros(3) if cs is None else np.asarray(cs, dtype=float).reshape(3)
        key = (float(self.rho), float(self.m), float(self.g), float(self.volume), float(self.zb), tuple(cur),
               tuple(float(getattr(self, a)) for a in _ATTR6),
               tuple(float(getattr(self, l)) for l in _LIN6), tuple(float(getattr(self, l + "_abs")) for l in _LIN6),
               float(self.Ix), float(self.Iy), float(self.Iz), float(self.xb), float(self.yb), self._extra_key())
        if key != self._pushed:
            p = self._params
            p.rho, p.m, p.g, p.volume = self.rho, self.m, self.g, self.volume
            p.xb, p.yb, p.zb = self.xb, self.yb, self.zb
            p.Ix, p.Iy, p.Iz = self.Ix, self.Iy, self.Iz
            for i, (a, l) in enumerate(zip(_ATTR6, _LIN6)):
                p.added_mass[i] = getattr(self, a)
                p.lin_damp[i] = getattr(self, l)
                p.quad_damp[i] = getattr(self, l + "_abs")
            for i in range(3):
                p.current[i] = cur[i]
            self._push_extra(p)
            self._ctx.set_params(p)
            self._pushed = key
        object.__setattr__(self, "_dirty", False)
        object.__setattr__(self, "_cs_pushed", csb)
        self._extra_mark_clean()

    def _rhs_single(self, x, u, dt, lag=None):
        """One dynamics() call through preallocated buffers and cached ctypes pointers (the per-call path is latency: the
        generic batched wrapper spends more time converting and allocating than the launch takes).  x, u: anything that
        reshapes to (nx,), (nu,) -- ValueError otherwise, like the reference's reshape.  lag: [8,3] array advanced in place."""
        io = self.__dict__.get("_io")
        if io is None:
            nx, nu = _lib.NX[self.MODEL], _lib.NU[self.MODEL]
            xb, ub, out = np.zeros(nx), np.zeros(nu), np.zeros(nx)
            io = (xb, ub, out, xb.ctypes.data, ub.ctypes.data, out.ctypes.data, nx, nu, self._ctx.lib.brov_rhs)
            object.__setattr__(self, "_io", io)
        xb, ub, out, xp, up, op, nx, nu, fn = io
        xb[...] = np.asarray(x, dtype=float).reshape(nx)
        ub[...] = np.asarray(u, dtype=float).reshape(nu)
        ctx = self._ctx
        if ctx._stream:
            ctx.set_stream(0)          # host path: back to the null stream (see Context.use_null_stream)
        rc = fn(ctx.h, self.MODEL, 1, xp, up, float(dt), lag.ctypes.data if lag is not None else None, op)
        if rc:
            ctx.check(rc, "brov_rhs")
        return out.copy()

    def _extra_clean(self):
        """Model-specific part of the fast check (the thruster model compares its geometry block)."""
        return True

    def _extra_mark_clean(self):
        pass

    def _push_extra(self, p):
        pass

    def _extra_key(self):
        """Model-specific part of the change key (the thruster model adds its geometry)."""
        return ()

    # ---- batched API (new; the reference only has the scalar dynamics()) --------------------
    def rollout(self, x0, U, dt, integrator="euler", lag=None, stride=1, lag_mode=_lib.LAG_PER_CALL):
        """simulate_physics for a batch: x0 [B,nx], U [B,T,nu] -> dict(traj [B,T//stride+1,nx], xT, lag)."""
        self._sync_params()
        return engine.rollout(self.MODEL, integrator, x0, U, dt, lag=lag, lag_mode=lag_mode, stride=stride, ctx=self._ctx)

    def simulate(self, x0, U_seq, dt, integrator="euler"):
        """One trajectory, the reference's simulate_physics signature: returns (len(U_seq)+1, nx).
        Starts from this object's current thruster-lag state and leaves it advanced, like the reference."""
        self._sync_params()
        lag = getattr(self, "_lag", None)
        r = engine.rollout(self.MODEL, integrator, np.asarray(x0, float)[None], np.asarray(U_seq, float)[None], dt,
                           lag=None if lag is None else lag[None], ctx=self._ctx)
        if lag is not None:
            self._lag[...] = r["lag"][0]
        return r["traj"][0]

    def one_step_rmse(self, X, U, dt):
        """one_step_rmse_physics (training/train_tank_brov2_koopmanEDMDc.py:237-247): Euler one-step predictions over a
        recording with ONE vehicle object (the lag runs through the whole sequence) == the H = 1 window evaluator."""
        return self.multistep_rmse_endpoint(X, U, 1, dt, "euler", carry_lag=True)

    def multistep_rmse_endpoint(self, X, U, H, dt, integrator="euler", carry_lag=True):
        """multistep_rmse_endpoint_physics (training/train_tank_brov2_full_comparison.py:469-487)."""
        self._sync_params()
        return engine.window_rmse(self.MODEL, integrator, X, U, H, dt, carry_lag=carry_lag, ctx=self._ctx)
