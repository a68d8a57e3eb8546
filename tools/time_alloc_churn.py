"""Does a hipFree / hipMalloc cycle make the next host -> device copy stall?  (tools/time_first_fit.py sees one 4.4 MB upload in two take
20-25 ms instead of 0.3 ms in the warm pass of a fit.)  Variants: copy into a buffer kept across iterations, copy into a fresh
allocation after freeing six buffers, with / without 20 ms of host work (numpy eigh) before the copy.
    python tools/time_alloc_churn.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from bluerov2_dynamics_amd import _lib  # noqa: E402

ctx = _lib.default_context(0)
lib = ctx.lib
X = np.random.default_rng(0).normal(size=(45823, 12))
G = np.random.default_rng(1).normal(size=(520, 520))
G = G @ G.T


def malloc(nbytes):
    p = ctypes.c_void_p()
    ctx.check(lib.brov_malloc(ctx.h, nbytes, ctypes.byref(p)), "malloc")
    return p


def throttled_ms():
    """cgroup v2: time this container's threads spent throttled by the CPU quota (cpu.max)"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1]) / 1e3
    except OSError:
        pass
    return float("nan")


def run(tag, churn, host_work, sleep):
    keep = malloc(X.nbytes)
    ts = []
    for rep in range(12):
        bufs = [malloc(s) for s in (4_400_000, 2_900_000, 48_000, 48_000, 2_200_000, 2_200_000)] if churn else []
        for b in bufs:
            ctx.check(lib.brov_memset(ctx.h, b, 0, 48_000), "memset")
        ctx.sync()
        for b in bufs:
            lib.brov_free(ctx.h, b)
        if host_work:
            np.linalg.eigh(G)
        if sleep:
            time.sleep(0.02)
        dst = malloc(X.nbytes) if churn else keep
        th0 = throttled_ms()
        t0 = time.perf_counter()
        ctx.check(lib.brov_memcpy_h2d(ctx.h, dst, X.ctypes.data, X.nbytes), "h2d")
        ts.append((time.perf_counter() - t0) * 1e3)
        if ts[-1] > 3.0:
            print(f"    stall of {ts[-1]:.1f} ms in rep {rep}: the container was throttled for {throttled_ms() - th0:.1f} ms meanwhile (cpu.stat)")
        if churn:
            lib.brov_free(ctx.h, dst)
    lib.brov_free(ctx.h, keep)
    print(f"{tag:58s} " + " ".join(f"{t:6.2f}" for t in ts))


from threadpoolctl import threadpool_info  # noqa: E402
print("BLAS pools:", [(t_["internal_api"], t_["num_threads"]) for t_ in threadpool_info()], " cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
run("kept buffer, no host work", False, False, False)
run("kept buffer, eigh(520) before the copy", False, True, False)
run("kept buffer, sleep 20 ms before the copy", False, False, True)
run("free six + fresh malloc, no host work", True, False, False)
run("free six + fresh malloc, eigh(520) before the copy", True, True, False)
run("free six + fresh malloc, sleep 20 ms before the copy", True, False, True)
