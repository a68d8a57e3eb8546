"""Where the first second goes: imports, context creation, first calls.  Run on the GPU box."""
import time
t0 = time.perf_counter()
import numpy as np
t1 = time.perf_counter(); print(f"import numpy            {t1 - t0:.3f} s")
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib
t2 = time.perf_counter(); print(f"import package (_lib)   {t2 - t1:.3f} s")
lib = _lib.load_library()
t3 = time.perf_counter(); print(f"load_library (+torch)   {t3 - t2:.3f} s")
ctx = _lib.Context(0)
t4 = time.perf_counter(); print(f"Context(0)              {t4 - t3:.3f} s")
from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2
rov = BlueROV2()
t5 = time.perf_counter(); print(f"BlueROV2()              {t5 - t4:.3f} s")
x = np.zeros(12); u = np.full(8, 0.1)
rov.dynamics(x, u, 0.02)
t6 = time.perf_counter(); print(f"first dynamics()        {t6 - t5:.3f} s")
rov.dynamics(x, u, 0.02)
t7 = time.perf_counter(); print(f"second dynamics()       {t7 - t6:.6f} s")
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
rng = np.random.default_rng(0)
X = np.cumsum(rng.normal(0, 0.02, (1600, 12)), 0); U = rng.uniform(-1, 1, (1600, 8))
m = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=500, gamma=3.0, ridge=0.1)
t8 = time.perf_counter(); m.fit(X, U); t9 = time.perf_counter(); print(f"first fit               {t9 - t8:.3f} s")
m.fit(X, U); t10 = time.perf_counter(); print(f"second fit              {t10 - t9:.3f} s")
