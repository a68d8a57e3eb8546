"""Host -> device upload of the recorded-shape operands (45 823 x 12 and x 8 doubles) through the C ABI, call by call: brov_malloc,
brov_memcpy_h2d (pageable source), brov_upload_bags (pinned staging blocks), brov_free -- in a torch-free process and with torch imported.
    BROV2_TORCH=auto|0|1 python tools/time_upload.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from bluerov2_dynamics_amd import _lib, engine  # noqa: E402

ctx = _lib.default_context(0)
lib = ctx.lib
rng = np.random.default_rng(0)
for N in (45823, 1_000_000):
    X = rng.normal(size=(N, 12))
    print(f"N = {N} ({X.nbytes / 1e6:.1f} MB), BROV2_TORCH={os.environ.get('BROV2_TORCH', 'auto')}, torch imported: {'torch' in sys.modules}")
    for rep in range(5):
        t0 = time.perf_counter()
        p = ctypes.c_void_p()
        ctx.check(lib.brov_malloc(ctx.h, X.nbytes, ctypes.byref(p)), "malloc")
        t1 = time.perf_counter()
        ctx.check(lib.brov_memcpy_h2d(ctx.h, p, X.ctypes.data, X.nbytes), "h2d")
        t2 = time.perf_counter()
        ptrs = np.array([X.ctypes.data], dtype=np.uint64)
        rows = np.array([N], dtype=np.int64)
        dst = np.zeros(1, dtype=np.int64)
        ctx.check(lib.brov_upload_bags(ctx.h, 1, ptrs.ctypes.data, rows.ctypes.data, dst.ctypes.data, 12, p), "bags")
        t3 = time.perf_counter()
        out = np.empty_like(X)
        ctx.check(lib.brov_memcpy_d2h(ctx.h, out.ctypes.data, p, X.nbytes), "d2h")
        t4 = time.perf_counter()
        ctx.check(lib.brov_free(ctx.h, p), "free")
        t5 = time.perf_counter()
        assert np.array_equal(out, X)
        print(f"  rep {rep}: malloc {1e3 * (t1 - t0):7.3f}  memcpy_h2d {1e3 * (t2 - t1):7.3f}  upload_bags {1e3 * (t3 - t2):7.3f}  memcpy_d2h {1e3 * (t4 - t3):7.3f}  "
              f"free {1e3 * (t5 - t4):7.3f} ms")
