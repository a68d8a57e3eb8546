"""Where the first KoopmanEDMDc.fit() of a fresh process goes, stage by stage, at the reference's recorded shape (N = 45 823, n = 12, r = 8,
500 RBFs, gamma = 3, ridge = 0.1: training/best_results.txt:3,761,798), then the same stages once more (warm).  Run on the GPU box:

    BROV2_TORCH=auto python tools/time_first_fit.py      # default: torch not imported, torch's libamdhip64 preloaded
    BROV2_TORCH=0    python tools/time_first_fit.py      # /opt/rocm's HIP runtime
    BROV2_TORCH=1    python tools/time_first_fit.py      # rounds 1-5: `import torch` first

Every stage ends with a device synchronisation, so the first-call column contains what that stage triggers lazily: the HIP runtime's
initialisation, the load of the code object of its translation unit (HIP defers it to the first launch from each .hip file), scratch
arenas, pinned blocks, the BLAS thread pool."""
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rows = []


def throttled_ms():
    """cgroup v2: thread-time this container spent throttled by its CPU quota (cpu.max) so far"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1]) / 1e3
    except OSError:
        pass
    return 0.0


_last = [throttled_ms(), sum(os.times()[:2])]


def lap(name, t_prev):
    t = time.perf_counter()
    th, cpu = throttled_ms(), sum(os.times()[:2])
    rows.append((name, t - t_prev, th - _last[0], cpu - _last[1]))
    _last[0], _last[1] = th, cpu
    return t


t = T0
import numpy as np  # noqa: E402
t = lap("import numpy", t)
from bluerov2_dynamics_amd import _lib, engine  # noqa: E402
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc  # noqa: E402,F401
t = lap("import bluerov2_dynamics_amd (_lib, engine, Koopman)", t)
lib = _lib.load_library()
t = lap(f"load_library: dlopen HIP runtime + libbrov2.so [{_lib.hip_runtime}]", t)
ctx = _lib.default_context(0)
t = lap("Context(0): hipInit, device properties, events, XCD probe launch", t)

N, n, r, k, gamma, ridge = 45823, 12, 8, 500, 3.0, 0.1
rng = np.random.default_rng(45823)
X = np.cumsum(rng.normal(0, 0.01, (N, n)), 0)
U = rng.uniform(-1, 1, (N, r))
t = lap("synthetic data on the host (not part of fit)", t)


KEEP = []
CACHE = {}


def stages(tag):
    d, p = n + k, n + k + r
    t = time.perf_counter()
    ns = engine._NativeArrays(ctx)
    Xd = engine.DevArray(ctx, X.shape)
    t = lap(f"{tag} brov_malloc X", t)
    if os.environ.get("FF_SLEEP_AFTER_MALLOC"):
        time.sleep(0.05)
        t = lap(f"{tag} sleep 50 ms", t)
    Xd.copy_from_host(X)
    t = lap(f"{tag} brov_memcpy_h2d X", t)
    Ud = ns.upload(U[:N - 1])
    ns.sync()
    t = lap(f"{tag} upload U (brov_malloc, brov_memcpy_h2d)", t)
    mean, var = engine.col_stats_dev(Xd, ctx=ctx)
    t = lap(f"{tag} column means / variances (colstats.hip)", t)
    C0, _ = engine.kmeanspp_dev(Xd, k, mean=mean, random_state=0, ctx=ctx)
    ns.sync()
    t = lap(f"{tag} k-means++ seeding (kmeans.hip + host draws)", t)
    C, inertia, iters = engine.kmeans_centers_dev(Xd, k, ctx=ctx)
    ns.sync()
    t = lap(f"{tag} centres again, whole call: stats + seeding + Lloyd ({iters} iterations; sortperm.hip)", t)
    GtG = ns.empty((p, p))
    engine.gram_dev(Xd, Ud, C, gamma, 1, N - 1, N, N - 1, GtG, None, ctx=ctx)
    Gh = ns.download(GtG)
    t = lap(f"{tag} lift + G^T G + download (edmdc.hip)", t)
    if os.environ.get("FF_NOLAPACK") and "P" in CACHE:
        P = CACHE["P"]
        time.sleep(0.015)
    else:
        with engine._blas_threads(p):
            P = engine._host_pinv(Gh, ridge, "auto")
        CACHE["P"] = P
    t = lap(f"{tag} host p x p solve (numpy LAPACK, pinv='auto')", t)
    M = ns.empty((p, d))
    engine.pinv_apply_dev(Xd, Ud, C, gamma, 1, N - 1, N, N - 1, P, M, ctx=ctx)
    Mh = ns.download(M)
    t = lap(f"{tag} (P G^T) Y apply + download", t)
    if os.environ.get("FF_KEEP"):
        KEEP.append((Xd, Ud, C0, C, GtG, M))
    del Xd, Ud, C0, C, GtG, M
    ns.sync()
    t = lap(f"{tag} free device buffers", t)
    return Mh


t1 = time.perf_counter()
stages("first:")
first_total = time.perf_counter() - t1
if os.environ.get("FF_SLEEP"):
    time.sleep(float(os.environ["FF_SLEEP"]))
    lap("sleep", t1)
if os.environ.get("FF_SYNC"):
    ctx.sync()
    lap("extra sync", t1)
t1 = time.perf_counter()
stages("warm: ")
warm_total = time.perf_counter() - t1
if os.environ.get("FF_NOLAPACK"):
    stages("warm2:")
    stages("warm3:")
m = KoopmanEDMDc(state_dim=n, input_dim=r, n_rbfs=k, gamma=gamma, ridge=ridge)
t1 = time.perf_counter()
m.fit(X, U)
lap("KoopmanEDMDc.fit(X, U) itself (warm process)", t1)
t1 = time.perf_counter()
sc = [m.multistep_rmse(X, U, H) for H in (1, 10, 100)]
lap("first multistep_rmse H = 1, 10, 100 (propagate.hip)", t1)

print(f"BROV2_TORCH={os.environ.get('BROV2_TORCH', 'auto')}  torch imported: {'torch' in sys.modules}  library: {os.path.getsize(_lib.library_path()) / 1e6:.2f} MB")
print("       wall   cpu(process)  throttled(container)")
for name, dt, th, cpu in rows:
    print(f"  {dt * 1e3:9.1f} ms {cpu * 1e3:8.1f} ms {th:8.1f} ms  {name}")
print(f"  first pass of the stages {first_total * 1e3:.1f} ms, warm pass {warm_total * 1e3:.1f} ms; "
      f"process start -> first pass done {sum(r_[1] for r_ in rows[:5 + 11]) * 1e3:.1f} ms (data generation included)")
