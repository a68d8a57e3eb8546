import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bluerov2_dynamics_amd import _lib
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
rng = np.random.default_rng(0)
X = np.cumsum(rng.normal(0, 0.02, (3000, 12)), 0); U = rng.uniform(-1, 1, (3000, 8))
m = KoopmanEDMDc(12, 8, n_rbfs=32, gamma=1.0, ridge=1e-2)
m.fit(X, U)
print("torch-free fit ok, torch imported:", "torch" in sys.modules, "|", _lib.hip_runtime)
t0 = time.perf_counter()
import torch
t1 = time.perf_counter()
print(f"late import torch: {t1 - t0:.2f} s, cuda available: {torch.cuda.is_available()}")
x = torch.ones(4, device="cuda"); torch.cuda.synchronize()
m2 = KoopmanEDMDc(12, 8, n_rbfs=32, gamma=1.0, ridge=1e-2, arrays="torch")
m2.fit(X, U)
print("torch-tensor fit after the late import equals the torch-free one:", np.array_equal(m.A_, m2.A_) and np.array_equal(m.B_, m2.B_), f"({time.perf_counter() - t1:.2f} s)")
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("HIP runtimes mapped:", maps)
