"""bluerov2_dynamics_amd -- MI355X-native batched BlueROV2 Fossen dynamics + Koopman EDMDc.

Drop-in modules (same class / method names as ViktorNfa/bluerov2_dynamics):
    bluerov2_dynamics_amd.fossen.BlueROV2          (thruster model, Euler angles)
    bluerov2_dynamics_amd.fossen.BlueROV2_thrust   (wrench input, Euler angles)
    bluerov2_dynamics_amd.fossen.BlueROV2_wrench   (wrench input, quaternion)
    bluerov2_dynamics_amd.Koopman.koopmanEDMDc     (KoopmanEDMDc)
Batched engine: bluerov2_dynamics_amd.engine; multi-GPU: bluerov2_dynamics_amd.dist.
Everything computes in hand-written HIP kernels behind the C ABI of include/brov2.h;
there is no CPU fallback.
"""
from ._lib import (BrovError, BrovParams, Context, default_context, default_params, discretise_lag, load_library, warm_up,
                   THRUSTER_EULER, WRENCH_EULER, WRENCH_QUAT, EULER, RK4, LAG_PER_CALL, LAG_PER_STEP,
                   LAYOUT_BTU, LAYOUT_TUB, LAYOUT_TPB)

__version__ = "0.1.0"
