import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
mode = sys.argv[1]
t0 = time.perf_counter()
if mode == "torch_first":
    import torch
    t1 = time.perf_counter(); print(f"import torch (first)        {t1 - t0:.3f} s")
from bluerov2_dynamics_amd import _lib
lib = _lib.load_library(); ctx = _lib.Context(0)
t2 = time.perf_counter(); print(f"library + context           {t2 - t0:.3f} s (cumulative)")
import torch
t3 = time.perf_counter(); print(f"import torch (now)          {t3 - t2:.3f} s")
x = torch.ones(4, device="cuda"); torch.cuda.synchronize()
t4 = time.perf_counter(); print(f"first cuda tensor           {t4 - t3:.3f} s")
print(f"total                       {t4 - t0:.3f} s")
