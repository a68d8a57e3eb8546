"""BlueROV2 Heavy with a 6-D body-wrench input and unit-quaternion attitude -- drop-in for the
reference's fossen/BlueROV2_wrench.py: class BlueROV2 (dynamics :322-367) plus the quaternion
helpers its callers import (training/train_tank_brov2_wrench_quat.py:20-21).  The helpers are
host-side conversions (NumPy); dynamics() and the rollouts run on the GPU."""
import numpy as np

from .. import _lib, engine
from ._vehicle import VehicleBase


def quat_normalize(q, eps=1e-12):
    """q / |q|, identity for a (near) zero quaternion (reference :27-36)."""
    q = np.asarray(q, dtype=float).reshape(4,)
    n = np.linalg.norm(q)
    return np.array([1.0, 0.0, 0.0, 0.0]) if n < eps else q / n


def quat_to_rotation_matrix(q):
    """Scalar-first quaternion -> R_{b->n} (reference :39-53)."""
    w, x, y, z = quat_normalize(q)
    return np.array([[1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - z * w), 2.0 * (x * z + y * w)],
                     [2.0 * (x * y + z * w), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - x * w)],
                     [2.0 * (x * z - y * w), 2.0 * (y * z + x * w), 1.0 - 2.0 * (x * x + y * y)]], dtype=float)


def quat_multiply(q1, q2):
    """Hamilton product (reference :56-68)."""
    a, b, c, d = np.asarray(q1, dtype=float).reshape(4,)
    e, f, g, h = np.asarray(q2, dtype=float).reshape(4,)
    return np.array([a * e - b * f - c * g - d * h, a * f + b * e + c * h - d * g,
                     a * g - b * h + c * e + d * f, a * h + b * g - c * f + d * e], dtype=float)


def quat_derivative(q, omega_body):
    """q_dot = 0.5 q (x) [0, omega] (reference :71-79)."""
    p, qr, r = np.asarray(omega_body, dtype=float).reshape(3,)
    return 0.5 * quat_multiply(q, np.array([0.0, p, qr, r]))


def euler_to_quat(phi, theta, psi):
    """Z-Y-X Euler angles -> quaternion (reference :86-106)."""
    c1, s1 = np.cos(float(phi) * 0.5), np.sin(float(phi) * 0.5)
    c2, s2 = np.cos(float(theta) * 0.5), np.sin(float(theta) * 0.5)
    c3, s3 = np.cos(float(psi) * 0.5), np.sin(float(psi) * 0.5)
    return quat_normalize([c3 * c2 * c1 + s3 * s2 * s1, c3 * c2 * s1 - s3 * s2 * c1,
                           c3 * s2 * c1 + s3 * c2 * s1, s3 * c2 * c1 - c3 * s2 * s1])


def quat_to_euler(q):
    """Quaternion -> (phi, theta, psi) (reference :109-132)."""
    w, x, y, z = quat_normalize(q)
    phi = np.arctan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    theta = np.arcsin(np.clip(2.0 * (w * y - z * x), -1.0, 1.0))
    psi = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return phi, theta, psi


def quat_to_yaw(q):
    w, x, y, z = quat_normalize(q)
    return float(np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z)))


class BlueROV2(VehicleBase):
    MODEL = _lib.WRENCH_QUAT

    def __init__(self, rho=1000.0, current_speed=None, device=None):
        cur = np.zeros(3, dtype=float) if current_speed is None else np.asarray(current_speed, dtype=float).reshape(3,)
        self._init_common(rho, cur, device)

    def dynamics(self, x, tau_body, dt=0.02):
        """xdot (13,) for x = [pos(3), q(4), nu(6)]; the quaternion is normalised on entry (:337)."""
        self._sync_params()
        return self._rhs_single(x, tau_body, 0.02)
