"""Scalar, per-call NumPy restatement of the thruster-model path IN THE REFERENCE'S SHAPE.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): a checker for the fixtures and the "what does the reference's way of computing cost on this
host" leg of bench.py's cpu_baseline -- never imported by the product.

oracle/brov2_oracle.c restates the same arithmetic as a tight C loop; this file keeps the reference's *cost structure*
instead: one Python call per right-hand side, small ndarray temporaries, a 6x6 Coriolis matrix filled entry by entry,
one np.cross and one 3x3 @ 3 lag update per thruster, four such calls per RK4 step.  That is what makes the reference
run at ~775 RK4 steps/s on one core, and bench.py times this restatement on the GPU box's host so the
figure beside the GPU number is measured there, not quoted.

Follows (paths relative to the reference checkout):
  fossen/BlueROV2.py:23-41   rotation_matrix            -> _rot
  fossen/BlueROV2.py:43-62   euler_kinematics_matrix    -> _j2      (cos(theta) clamp incl. sign(0) = 0)
  fossen/BlueROV2.py:79-157  constants, :172-232 thruster placements (angles as printed there)
  fossen/BlueROV2.py:245-263 thrust polynomial + lag, :265-278 allocation by cross products
  fossen/BlueROV2.py:280-355 Coriolis (with the author's sign choice at :293,297), damping, restoring
  fossen/BlueROV2.py:357-400 dynamics() -- advances the eight lag filters on every call
  fossen/BlueROV2.py:464-510 ThrusterLag: ZOH via scipy.signal.cont2discrete, x <- Ad x + Bd u, y = Cc x
  training/train_tank_brov2_rk4.py:375-396 RK4 loop, training/train_tank_brov2_full_comparison.py:453-466 Euler loop
"""
import numpy as np
from scipy.signal import cont2discrete

_AC = np.array([[-89.0, -72.33, -26.54], [128.0, 0.0, 0.0], [0.0, 32.0, 0.0]])
_BC = np.array([[8.0], [0.0], [0.0]])
_CC = np.array([[0.0, 5.992, 3.317]])
_DC = np.zeros((1, 1))


def _rot(phi, theta, psi):
    cf, sf = np.cos(phi), np.sin(phi)
    ct, st = np.cos(theta), np.sin(theta)
    cp, sp = np.cos(psi), np.sin(psi)
    return np.array([[cp * ct, -sp * cf + cp * st * sf, sp * sf + cp * cf * st],
                     [sp * ct, cp * cf + sf * st * sp, -cp * sf + st * sp * cf],
                     [-st, ct * sf, ct * cf]], dtype=float)


def _j2(phi, theta, eps=1e-7):
    sf, cf = np.sin(phi), np.cos(phi)
    st, ct = np.sin(theta), np.cos(theta)
    if abs(ct) < eps:
        ct = eps * np.sign(ct)
    tt = st / ct
    return np.array([[1.0, sf * tt, cf * tt], [0.0, cf, -sf], [0.0, sf / ct, cf / ct]], dtype=float)


def _rz(a):
    s, c = np.sin(a), np.cos(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class _Lag:
    """One thruster's third-order lag (stateful)."""

    def __init__(self):
        self.dt = None
        self.Ad = self.Bd = None
        self.x = np.zeros(3)

    def step(self, u, dt):
        if self.dt != dt:
            self.Ad, self.Bd, _, _, _ = cont2discrete((_AC, _BC, _CC, _DC), dt, method="zoh")
            self.dt = dt
        self.x = self.Ad @ self.x + self.Bd[:, 0] * u
        return float((_CC @ self.x)[0])


class ScalarBlueROV2:
    """Same numbers as the reference's fossen.BlueROV2.BlueROV2 with default arguments, computed the same way."""

    def __init__(self):
        self.m, self.g, rho, vol = 13.5, 9.82, 1000.0, 0.0134
        self.W, self.B = self.m * self.g, rho * self.g * vol
        self.xb, self.yb, self.zb = 0.0, 0.0, -0.01
        self.Ix, self.Iy, self.Iz = 0.26, 0.23, 0.37
        self.am = (-6.36, -7.12, -18.68, -0.189, -0.135, -0.222)          # Xu_dot .. Nr_dot
        self.dl = (-13.7, -0.0, -33.0, -0.0, -0.8, -0.0)                  # Xu .. Nr
        self.dq = (-141.0, -217.0, -190.0, -1.19, -0.47, -1.5)            # Xu|u| .. Nr|r|
        M = np.diag([self.m, self.m, self.m, self.Ix, self.Iy, self.Iz]) + np.diag([-a for a in self.am])
        self.Minv = np.linalg.inv(M)
        self.cur = np.zeros(3)
        r14, r58 = np.array([0.156, 0.111, 0.085]), np.array([0.12, 0.218, 0.0])
        e = np.array([1.0 / np.sqrt(2), -1.0 / np.sqrt(2), 0.0])
        ang_r = (0.0, 5.05, 1.91, np.pi, 0.0, 4.15, 1.01, np.pi)
        ang_e = (0.0, np.pi / 2, 3 * np.pi / 2, np.pi)
        self.thr = []
        for i in range(8):
            r = _rz(ang_r[i]) @ (r14 if i < 4 else r58)
            d = _rz(ang_e[i]) @ e if i < 4 else np.array([0.0, 0.0, -1.0])
            self.thr.append((r, d))
        self.lags = [_Lag() for _ in range(8)]

    def _thrust(self, V, i, dt):
        V3, V5, V7, V9 = V ** 3, V ** 5, V ** 7, V ** 9
        F = -140.3 * V9 + 389.9 * V7 - 404.1 * V5 + 176.0 * V3 + 8.9 * V
        return float(self.lags[i].step(F, dt))

    def _tau(self, u, dt):
        tau = np.zeros(6)
        for i in range(8):
            r, d = self.thr[i]
            f = self._thrust(u[i], i, dt) * d
            tau[0:3] += f
            tau[3:6] += np.cross(r, f)
        return tau

    def _coriolis(self, nu):
        u, v, w, p, q, r = nu
        m, Ix, Iy, Iz = self.m, self.Ix, self.Iy, self.Iz
        Xu, Yv, Zw, Kp, Mq, Nr = self.am
        C = np.zeros((6, 6))
        A = np.zeros((6, 6))
        C[0, 4], C[0, 5], C[1, 3], C[1, 5], C[2, 3], C[2, 4] = m * w, -m * v, -m * w, m * u, m * v, -m * u
        C[3, 1], C[3, 2], C[3, 4], C[3, 5] = m * w, -m * v, Iz * r, -Iy * q
        C[4, 0], C[4, 2], C[4, 3], C[4, 5] = -m * w, m * u, -Iz * r, Ix * p
        C[5, 0], C[5, 1], C[5, 3], C[5, 4] = m * v, -m * u, Iy * q, -Ix * p
        A[0, 4], A[0, 5], A[1, 3], A[1, 5], A[2, 3], A[2, 4] = -Zw * w, Yv * v, Zw * w, -Xu * u, -Yv * v, Xu * u
        A[3, 1], A[3, 2], A[3, 4], A[3, 5] = -Zw * w, Yv * v, -Nr * r, Mq * q
        A[4, 0], A[4, 2], A[4, 3], A[4, 5] = Zw * w, -Xu * u, Nr * r, -Kp * p
        A[5, 0], A[5, 1], A[5, 3], A[5, 4] = -Yv * v, Xu * u, -Mq * q, Kp * p
        return C + A

    def _damping(self, nur):
        D = np.zeros((6, 6))
        for i in range(6):
            D[i, i] = -self.dl[i] - self.dq[i] * abs(nur[i])
        return D

    def _restoring(self, phi, theta):
        wb = self.W - self.B
        sf, cf, st, ct = np.sin(phi), np.cos(phi), np.sin(theta), np.cos(theta)
        g = np.zeros(6)
        g[0], g[1], g[2] = wb * st, -wb * ct * sf, -wb * ct * cf
        g[3] = (self.yb * self.B) * ct * cf - (self.zb * self.B) * ct * sf
        g[4] = -(self.zb * self.B) * st - (self.xb * self.B) * ct * cf
        g[5] = (self.xb * self.B) * ct * sf + (self.yb * self.B) * st
        return g

    def dynamics(self, x, u, dt):
        eta, nu = x[0:6], x[6:12]
        phi, theta, psi = eta[3:6]
        R = _rot(phi, theta, psi)
        J = _j2(phi, theta)
        nur = np.copy(nu)
        nur[:3] -= R.T.dot(self.cur)
        C = self._coriolis(nu)
        D = self._damping(nur)
        g = self._restoring(phi, theta)
        tau = np.copy(self._tau(u, dt))
        nud = self.Minv.dot(tau - C.dot(nu) - D.dot(nur) - g)
        return np.concatenate([np.concatenate([R.dot(nu[0:3]), J.dot(nu[3:6])]), nud])


def simulate(x0, U, dt, integrator="rk4", rov=None):
    """simulate_physics of the training scripts: returns (len(U)+1, 12) including x0."""
    rov = rov or ScalarBlueROV2()
    x = np.array(x0, dtype=float).copy()
    out = [x.copy()]
    for k in range(len(U)):
        if integrator == "euler":
            x = x + dt * rov.dynamics(x, U[k], dt)
        else:
            k1 = rov.dynamics(x, U[k], dt)
            k2 = rov.dynamics(x + 0.5 * dt * k1, U[k], dt)
            k3 = rov.dynamics(x + 0.5 * dt * k2, U[k], dt)
            k4 = rov.dynamics(x + dt * k3, U[k], dt)
            x = x + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append(x.copy())
    return np.stack(out)
