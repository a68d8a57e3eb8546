"""BlueROV2 Heavy with a direct 6-D body-wrench input and Euler-angle state -- drop-in for the
reference's fossen/BlueROV2_thrust.py (stateless dynamics(x, tau_body, dt=0.02), :235-282)."""
import numpy as np

from .. import _lib, engine
from ._vehicle import VehicleBase
from .BlueROV2 import euler_kinematics_matrix, rotation_matrix  # noqa: F401  (same helpers as the reference module)


class BlueROV2(VehicleBase):
    MODEL = _lib.WRENCH_EULER

    def __init__(self, rho=1000.0, current_speed=None, device=None):
        cur = np.zeros(3, dtype=float) if current_speed is None else np.asarray(current_speed, dtype=float).reshape(3,)
        self._init_common(rho, cur, device)

    def dynamics(self, x, tau_body, dt=0.02):
        """xdot (12,); raises ValueError on wrongly sized inputs like the reference's reshape (:245-246)."""
        self._sync_params()
        return self._rhs_single(x, tau_body, 0.02)
