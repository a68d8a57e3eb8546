#!/usr/bin/env python3
"""In-kernel clock of the config-2 rollout launch (MI355X_MICROARCH.md, DVFS give-back item 6): needs a diagnostic build of
the library with -DBROV_CLOCK_STAMPS=1 (tools/build_variants.py clock=-DBROV_CLOCK_STAMPS=1), selected with BROV2_LIBRARY.

    BROV2_LIBRARY=$PWD/build_variants/clock/libbrov2.so python tools/clock_probe.py            # two-wave kernel
    BROV2_ROLLOUT_SINGLE_LANE=1 BROV2_LIBRARY=... python tools/clock_probe.py                   # one-lane kernel
Back-to-back launches for ~2 s first (the clock settles), then the stamps of the last launch: shader cycles and 100 MHz ticks
around the time loop of every workgroup; clock = cycles / ticks x 100 MHz (median over workgroups)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bluerov2_dynamics_amd import _lib, engine

B, T, dt = 65536, 5000, 0.02
dev = torch.device("cuda")
ctx = _lib.default_context(0)
U = torch.empty((T, 4, B, 2), dtype=torch.float64, device=dev)
engine.fill_controls_dev(U, "tpb", "iid", seed=0x5EED, T_total=T, ctx=ctx)
x0 = torch.zeros((B, 12), dtype=torch.float64, device=dev); x0[:, 2] = 5.0
traj = torch.empty((T + 1, 6, B, 2), dtype=torch.float64, device=dev)
xT = torch.empty((B, 12), dtype=torch.float64, device=dev)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < float(os.environ.get("BROV2_CLOCK_WARM_S", "3")):
    engine.rollout_dev(_lib.THRUSTER_EULER, "rk4", x0, U, dt, traj=traj, xT=xT, layout="tpb", ctx=ctx); n += 1
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); engine.rollout_dev(_lib.THRUSTER_EULER, "rk4", x0, U, dt, traj=traj, xT=xT, layout="tpb", ctx=ctx); e1.record()
torch.cuda.synchronize()
nblk = 256
buf = (ctypes.c_ulonglong * (4 * nblk))()
fn = ctx.lib.brov_debug_clock_stamps
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(ctypes.addressof(buf), nblk) == 0
s = np.array(buf, dtype=np.uint64).reshape(nblk, 4).astype(np.float64)
cyc, ticks = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
ghz = cyc / ticks * 0.1
print("%s: kernel %.3f ms after %d warm launches; time loop %.2f M shader cycles in %.3f ms -> in-kernel clock median %.3f GHz (min %.3f, max %.3f)" % (
    "one-lane" if os.environ.get("BROV2_ROLLOUT_SINGLE_LANE") == "1" else "two-wave", e0.elapsed_time(e1), n,
    np.median(cyc) / 1e6, np.median(ticks) / 1e5, np.median(ghz), ghz.min(), ghz.max()))
