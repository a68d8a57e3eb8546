"""Host-only entry points of libbrov2.so over a sweep of shapes -- run under the sanitizer build (tools/sanitize_host.sh).
Nothing here needs a GPU: task tables / dynamic programmes of the Gram and apply passes, the W-rows plan, derive_fast through
brov_get_derived, the Pade-13 discretisation, and the argument validation of every entry point that takes a context."""
import ctypes
import itertools
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bluerov2_dynamics_amd import _lib

lib = _lib.load_library()
n_dec = 0
for n, r in ((12, 8), (12, 6), (13, 6), (5, 2), (1, 0), (16, 64), (9, 4), (3, 1)):
    for k in list(range(1, 70)) + [80, 96, 100, 128, 160, 200, 255, 256, 257, 300, 400, 500, 511, 512, 513, 530, 600, 700, 1000, 1024, 1500, 2048, 4096]:
        nt, ns = ctypes.c_int(0), ctypes.c_int(0)
        for fn in (lib.edmdc_gram_decomposition, lib.edmdc_gtg_decomposition):
            rc = fn(n, r, k, ctypes.byref(nt), ctypes.byref(ns))
            assert rc == 0 and nt.value > 0 and ns.value > 0, (fn.__name__, n, r, k, rc, nt.value, ns.value)
        a = [ctypes.c_int(0) for _ in range(4)]
        rc = lib.edmdc_apply_decomposition(n, r, k, *[ctypes.byref(x) for x in a])
        assert rc == 0 and all(x.value > 0 for x in a), ("apply", n, r, k, rc, [x.value for x in a])
        n_dec += 1
# out-of-range shapes are refused, not walked
for n, r, k in ((0, 1, 1), (17, 1, 1), (12, 65, 8), (12, 8, 0), (12, 8, -3), (12, 8, 65535 * 16 + 1)):
    nt, ns = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.edmdc_gram_decomposition(n, r, k, ctypes.byref(nt), ctypes.byref(ns)) != 0, (n, r, k)
print(f"decompositions: {n_dec} shapes x 3 tables ok")

# Pade discretisation + derived constants over dt and perturbed parameter sets
rng = np.random.default_rng(0)
n_par = 0
for dt in (1e-4, 0.01, 0.02, 0.05, 0.5, 5.0):
    Ad, Bd = _lib.discretise_lag(dt)
    assert np.all(np.isfinite(Ad)) and np.all(np.isfinite(Bd))
for _ in range(200):
    p = _lib.default_params()
    p.m *= float(rng.uniform(0.5, 2)); p.zb = float(rng.normal(0, 0.05)); p.xb = float(rng.normal(0, 0.01))
    for i in range(6):
        p.added_mass[i] *= float(rng.uniform(0.5, 2))
    for i in range(8):
        for j in range(3):
            p.thr_r[i][j] += float(rng.normal(0, 0.01))
    for i in range(9):
        p.lag_Ac[i] *= float(rng.uniform(0.8, 1.2))
    Minv, T = _lib.derived(p)
    Ad, Bd = _lib.discretise_lag(float(rng.choice([0.01, 0.02, 0.05])), p)
    assert np.all(np.isfinite(Minv)) and np.all(np.isfinite(T)) and np.all(np.isfinite(Ad))
    n_par += 1
print(f"derived / discretise: {n_par} parameter sets ok")

# every entry point with a ctx argument must refuse a NULL ctx (and bad pointers) with a status, without touching memory
h = ctypes.c_void_p()
rc = lib.brov_create(0, ctypes.byref(h))
print("brov_create without a GPU ->", _lib.STATUS.get(rc, rc))
null = ctypes.c_void_p(None)
calls = 0
for name, (res, args) in _lib.SIGNATURES.items():
    if res is not ctypes.c_int or not args or args[0] is not _lib.c_void_p or name in ("brov_comm_unique_id",):
        continue
    if name.startswith("brov_comm") or name == "edmdc_gram_allreduce_dev":
        continue
    vals = []
    for a in args:
        if a in (ctypes.c_int, _lib.i64, ctypes.c_uint64, ctypes.c_size_t):
            vals.append(a(0))
        elif a is ctypes.c_double:
            vals.append(a(0.0))
        else:
            vals.append(None)
    rc = getattr(lib, name)(*vals)
    assert isinstance(rc, int), name
    calls += 1
assert lib.edmdc_kmeans_relocations(None) == 0
print(f"NULL-context calls: {calls} entry points returned a status")

# round 5: the library's far-row selection (NumPy's introselect restated, host only): the fixture's cases and random ones with ties / NaNs,
# the result checked for being the n_empty largest
import os
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "farselect.npz"))
n_sel = 0
cases = [(np.ascontiguousarray(g[k[:-2] + "_d"]), int(g[k[:-2] + "_n"])) for k in g.files if k.endswith("_d")]
for _ in range(300):
    N = int(rng.integers(1, 5000))
    d = rng.random(N)
    if rng.random() < 0.4:
        d = np.round(d, int(rng.integers(0, 3)))
    if rng.random() < 0.2:
        d[rng.integers(0, N, max(1, N // 50))] = np.nan
    cases.append((d, int(rng.integers(1, min(N, 70) + 1))))
for d, ne in cases:
    out = np.empty(ne, dtype=np.int64)
    assert lib.edmdc_far_select_numpy(d.ctypes.data, len(d), ne, out.ctypes.data) == 0
    key = np.where(np.isnan(d), np.inf, d)
    assert len(set(out.tolist())) == ne and np.sort(key[out])[0] >= np.sort(key)[-ne]
    n_sel += 1
print(f"far-row selection: {n_sel} cases ok")
print("sanitize sweep: ok")
