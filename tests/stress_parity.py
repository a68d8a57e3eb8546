#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (not part of the test suite: minutes, not seconds): random shapes, layouts,
strides, bag structures and chunk sizes through the C ABI against the oracle.  Prints the worst relative error per family
and exits non-zero on the first violation.

    python tests/stress_parity.py [seconds per family = 40] [seed = 0] [families, comma separated: copies,linear_multistep,applies,lloyds,lloyds_list,rollouts,windows,grams,multistep,kmeanspp]

Lives under tests/ because it checks against oracle/ (test infrastructure); tests/test_gpu_parity.py runs a short sweep."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bluerov2_dynamics_amd import _lib, engine
from oracle import fossen_c, edmdc_numpy as ek

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NX, NU = _lib.NX, _lib.NU


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def rollouts():
    worst, n, t0 = 0.0, 0, time.time()
    while time.time() - t0 < budget:
        model = int(rng.integers(0, 3))
        integ = str(rng.choice(["euler", "rk4"]))
        B = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 700]))
        T = int(rng.choice([1, 2, 3, 17, 64, 65, 130, 257]))
        stride = int(rng.choice([1, 1, 2, 5, T]))
        layout = str(rng.choice(["btu", "tub", "tpb"]))
        lag_mode = int(rng.integers(0, 2))
        nx, nu = NX[model], NU[model]
        x0 = rng.normal(0, 0.3, (B, nx))
        if nx == 13:
            x0[:, 3:7] /= np.linalg.norm(x0[:, 3:7], axis=1, keepdims=True)
        U = rng.uniform(-1, 1, (B, T, nu)) * (1.0 if model == 0 else 8.0)
        lag = rng.normal(0, 0.2, (B, 8, 3)) if (model == 0 and rng.random() < 0.5) else None
        ref = fossen_c.rollout(model, 0 if integ == "euler" else 1, x0, U, 0.02, lag=lag, lag_mode=lag_mode, sub=stride)
        if layout == "btu":
            Ud = U
        elif layout == "tub":
            Ud = np.ascontiguousarray(U.transpose(1, 2, 0))
        else:                                                     # [T][nu/2][B][2]
            Ud = np.ascontiguousarray(U.reshape(B, T, nu // 2, 2).transpose(1, 2, 0, 3))
        got = engine.rollout(model, integ, x0, Ud, 0.02, lag=lag, lag_mode=lag_mode, layout=layout, stride=stride)
        if layout == "btu":
            tr = got["traj"]
        elif layout == "tub":
            tr = got["traj"].transpose(2, 0, 1)
        else:                                                     # [rows][ceil(nx/2)][B][2]
            tr = got["traj"].transpose(2, 0, 1, 3).reshape(B, got["traj"].shape[0], -1)[:, :, :nx]
        # lanes that pass within 0.05 rad of the Euler-angle singularity amplify the last bit by 1 / cos(theta)^2 (up to the
        # reference's 1e-7 clamp) in BOTH implementations: there only finiteness is compared, everywhere else 1e-9
        scale = np.abs(ref["traj"][np.isfinite(ref["traj"])]).max()
        lane_err = np.nan_to_num(np.abs(tr - ref["traj"]), nan=0.0, posinf=0.0).max(axis=(1, 2)) / scale
        if nx == 12:
            full = ref if stride == 1 else fossen_c.rollout(model, 0 if integ == "euler" else 1, x0, U, 0.02, lag=lag, lag_mode=lag_mode, sub=1)
            near = np.abs(np.cos(full["traj"][:, :, 4])).min(axis=1) < 0.05
        else:
            near = np.zeros(B, dtype=bool)
        assert np.array_equal(np.isfinite(tr), np.isfinite(ref["traj"])), ("rollout finiteness", model, integ, B, T)
        keep = ~near
        e = float(lane_err[keep].max()) if keep.any() else 0.0
        e = max(e, relerr(got["xT"][keep], ref["xT"][keep]) if keep.any() else 0.0)
        if model == 0 and got["lag"] is not None:
            e = max(e, relerr(got["lag"], ref["lag"]))
        tol = 1e-9
        if not (np.isfinite(e) and e < tol):
            err = np.abs(tr - ref["traj"])
            per_lane = err.max(axis=(1, 2)); b = int(per_lane.argmax())
            print("  worst lane", b, "abs err", err.max(), "lanes above 1e-11:", int((per_lane > 1e-11).sum()), "of", B,
                  "median lane err", float(np.median(per_lane)))
            print("  err over time (every 16 rows):", err[b].max(axis=1)[::16])
            print("  theta of that lane: min %.4f max %.4f; max |state| %.3f" % (ref["traj"][b, :, 4].min(), ref["traj"][b, :, 4].max(), np.abs(ref["traj"][b]).max()))
        assert np.isfinite(e) and e < tol, ("rollout", model, integ, B, T, stride, layout, lag_mode, e)
        worst, n = max(worst, e), n + 1
    print(f"rollouts   : {n} cases, worst rel err {worst:.2e}", flush=True)


def grams():
    worst, n, t0 = 0.0, 0, time.time()
    ctx = _lib.default_context()
    while time.time() - t0 < budget:
        n_, r = [(12, 8), (12, 6), (13, 6), (9, 4), (5, 2), (13, 8), (12, 10)][int(rng.integers(0, 7))]
        k = int(rng.choice([1, 7, 16, 17, 48, 100, 200, 257, 500, 512]))
        nb = int(rng.integers(1, 5)) if rng.random() < 0.6 else int(rng.integers(5, 300))       # (round 5: long ragged lists, bags without a pair)
        lens = [int(rng.choice([0, 1, 2, 3, 5, 33, 100, 257, 1000] if nb > 4 else [2, 3, 5, 33, 100, 257, 1000])) for _ in range(nb)]
        if max(lens) < 2:
            lens[0] = 5
        chunk = int(rng.choice([64, 100, 257, 4096, 1 << 20]))
        ctx.check(ctx.lib.edmdc_set_chunk_rows(ctx.h, chunk), "edmdc_set_chunk_rows")
        Xs = [np.cumsum(rng.normal(0, 0.05, (m, n_)), 0) for m in lens]
        short = rng.random() < 0.3                      # inputs one row shorter than the states (all the reference reads)
        Us = [rng.uniform(-1, 1, (m, r)) for m in lens]
        C = rng.normal(0, 0.4, (k, n_))
        g = float(rng.choice([0.3, 1.0, 3.0]))
        GtG, GtY, npairs = engine.gram(Xs, [u[:max(len(u) - 1, 0)] for u in Us] if short else Us, C, g)
        Go, Yo, npo = ek.gram(Xs, Us, C, g)
        e = max(np.abs(GtG - Go).max() / np.abs(Go).max(), np.abs(GtY - Yo).max() / max(1e-300, np.abs(Yo).max()))
        assert npairs == npo and np.isfinite(e) and e < 1e-11, ("gram", n_, r, k, lens, chunk, e)
        worst, n = max(worst, float(e)), n + 1
    ctx.check(ctx.lib.edmdc_set_chunk_rows(ctx.h, 1 << 20), "edmdc_set_chunk_rows")
    print(f"Gram       : {n} cases, worst rel err {worst:.2e}", flush=True)


def multistep():
    worst, n, t0 = 0.0, 0, time.time()
    while time.time() - t0 < budget:
        n_, r = [(12, 8), (13, 6), (9, 4)][int(rng.integers(0, 3))]
        k = int(rng.choice([8, 40, 100, 200]))
        N = int(rng.choice([30, 129, 300, 1500]))
        H = int(rng.choice([1, 2, 10, 29]))
        if H >= N:
            continue
        X = np.cumsum(rng.normal(0, 0.02, (N, n_)), 0)
        U = rng.uniform(-1, 1, (N, r))
        C = X[rng.choice(N, k, replace=k > N)]
        A, B_ = ek.fit([X], [U], C, 1.0, 1e-2)
        se, _ = engine.multistep_se(X, U, C, 1.0, A, B_, H)
        ref = ek.multistep_rmse(X, U, C, 1.0, A, B_, H)
        got = np.sqrt(se / ((N - H) * n_))
        e = abs(got - ref) / max(ref, 1e-300)
        assert np.isfinite(e) and e < 1e-9, ("multistep", n_, r, k, N, H, got, ref)
        worst, n = max(worst, float(e)), n + 1
    print(f"multistep  : {n} cases, worst rel err {worst:.2e}", flush=True)


def windows():
    worst, n, t0 = 0.0, 0, time.time()
    while time.time() - t0 < budget:
        model = int(rng.integers(0, 6))                           # incl. the double-integrator baselines (3..5)
        integ = str(rng.choice(["euler", "rk4"]))
        N = int(rng.choice([12, 70, 200, 1000, 4200]))
        H = int(rng.choice([1, 2, 10, 33]))
        if H >= N:
            continue
        nx, nu = NX[model], NU[model]
        X = np.cumsum(rng.normal(0, 0.01, (N, nx)), 0)
        X[:, 3:6] *= 0.2
        if nx == 13:
            X[:, 3:7] = rng.normal(0, 1, (N, 4)); X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
        U = rng.uniform(-1, 1, (N, nu)) * (1.0 if nu == 8 else 5.0)
        carry = bool(rng.integers(0, 2))
        if model >= 3:
            Kl, Ka = rng.normal(0, 0.5, (nu, 3)), rng.normal(0, 0.5, (nu, 3))
            fossen_c.set_di_gains(Kl, Ka)
            _lib.default_context().set_di_gains(Kl, Ka)
        se_ref, per_ref = fossen_c.window_endpoint_se(model, 0 if integ == "euler" else 1, X, U, H, 0.02, carry_lag=carry)
        se, per = engine.window_endpoint_se(model, integ, X, U, H, 0.02, carry_lag=carry)
        e = max(abs(se - se_ref) / max(se_ref, 1e-300), float(np.max(np.abs(per - per_ref)) / max(1e-300, np.max(np.abs(per_ref)))))
        assert np.isfinite(e) and e < 1e-9, ("window", model, integ, N, H, carry, e)
        worst, n = max(worst, e), n + 1
    print(f"windows    : {n} cases, worst rel err {worst:.2e}", flush=True)


def kmeanspp():
    """Seed indices equal to scikit-learn's -- except where two candidates of a round tie in exact arithmetic (two mutually
    nearest uncovered points both drawn: pot - closest[a] - closest[b] + d(a, b) either way), which happens at tiny N with
    k ~ N / 3.  There the winner is decided by the summation order of the potentials (scikit-learn's own comes out of a BLAS
    matrix-vector product), so a mismatch is accepted iff the two choices' potentials agree to 1e-10 at the first differing
    round (replayed here with NumPy)."""
    import torch
    from sklearn.cluster import kmeans_plusplus
    from sklearn.metrics.pairwise import euclidean_distances
    from sklearn.utils.extmath import row_norms, stable_cumsum
    ties, n, t0 = 0, 0, time.time()
    while time.time() - t0 < budget:
        N = int(rng.choice([50, 4095, 4096, 4097, 9000, 20000, 50000]))
        n_ = int(rng.choice([5, 12, 13]))
        k = int(rng.choice([2, 3, 16, 64, 200]))
        if k > N:
            continue
        X = np.concatenate([rng.normal(m, 0.5, (N // 4 + 1, n_)) for m in rng.uniform(-2, 2, (4, n_))])[:N]
        seed = int(rng.integers(0, 1000))
        mean = X.mean(0)
        Xc = X - mean
        C_ref, idx_ref = kmeans_plusplus(Xc, k, random_state=np.random.RandomState(seed))
        C, idx = engine.kmeanspp_dev(torch.from_numpy(X).cuda(), k, mean=mean, random_state=seed)
        n += 1
        if np.array_equal(idx, idx_ref):
            continue
        j = int(np.argmax(idx != idx_ref))
        rs = np.random.RandomState(seed)
        xsq = row_norms(Xc, squared=True)
        first = rs.choice(N, p=np.full(N, 1.0 / N))
        closest = euclidean_distances(Xc[first, None], Xc, Y_norm_squared=xsq, squared=True)[0]
        pot = closest.sum()
        L = 2 + int(np.log(k))
        for c in range(1, j + 1):
            cand = np.searchsorted(stable_cumsum(closest), rs.uniform(size=L) * pot)
            np.clip(cand, None, N - 1, out=cand)
            d = euclidean_distances(Xc[cand], Xc, Y_norm_squared=xsq, squared=True)
            np.minimum(closest, d, out=d)
            pots = d.sum(axis=1)
            if c == j:
                assert idx[j] in cand and idx_ref[j] in cand, ("kmeans++ pick is not a candidate", N, n_, k, seed, j)
                pa, pb = pots[list(cand).index(idx[j])], pots[list(cand).index(idx_ref[j])]
                assert abs(pa - pb) <= 1e-10 * abs(pb), ("kmeans++ differs without a tie", N, n_, k, seed, j, pa, pb)
            best = list(cand).index(idx_ref[c])
            pot, closest = pots[best], d[best]
        ties += 1
    print(f"k-means++  : {n} cases, seed indices equal to scikit-learn's except {ties} exact-arithmetic ties", flush=True)


def applies():
    """edmdc_pinv_apply (round 3: tuned W-rows kernel with both block orientations + the W^T Y task packing by DP) against NumPy's
    (P G^T) Y and against the plain W-rows kernel, random shapes / bags / chunk sizes."""
    worst, n, t0 = 0.0, 0, time.time()
    ctx = _lib.default_context()
    while time.time() - t0 < budget:
        n_, r = [(12, 8), (12, 6), (13, 6), (9, 4), (5, 2), (13, 8)][int(rng.integers(0, 6))]
        k = int(rng.choice([1, 7, 16, 17, 48, 64, 80, 96, 100, 160, 200, 257, 500, 512, 530, 600]))     # every nt mod 6 of the W-rows plan
        nb = int(rng.integers(1, 4))
        lens = [int(rng.choice([2, 3, 33, 100, 193, 257, 1000])) for _ in range(nb)]
        chunk = int(rng.choice([64, 100, 192, 257, 4096, 1 << 20]))
        ctx.check(ctx.lib.edmdc_set_chunk_rows(ctx.h, chunk), "edmdc_set_chunk_rows")
        Xs = [np.cumsum(rng.normal(0, 0.05, (m, n_)), 0) for m in lens]
        Us = [rng.uniform(-1, 1, (m, r)) for m in lens]
        C = rng.normal(0, 0.4, (k, n_))
        g = float(rng.choice([0.3, 1.0, 3.0]))
        p = n_ + k + r
        P = rng.normal(0, 1, (p, p)) / np.sqrt(p)
        got = []
        for v in (0, 1):
            ctx.set_apply_variant(v)
            got.append(engine.pinv_apply(Xs, Us, C, g, P))
        ctx.set_apply_variant(0)
        Mo = np.zeros_like(got[0])
        for Xb, Ub in zip(Xs, Us):
            G = np.hstack([ek.lift(Xb[:-1], C, g), Ub[:-1]])
            Mo += (P @ G.T) @ ek.lift(Xb[1:], C, g)
        sc = max(1e-300, np.abs(Mo).max())
        e = max(np.abs(got[0] - Mo).max() / sc, np.abs(got[0] - got[1]).max() / sc)
        assert np.isfinite(e) and e < 1e-11, ("apply", n_, r, k, lens, chunk, e)
        worst, n = max(worst, float(e)), n + 1
    ctx.check(ctx.lib.edmdc_set_chunk_rows(ctx.h, 1 << 20), "edmdc_set_chunk_rows")
    print(f"fit() apply: {n} cases, worst rel err {worst:.2e}", flush=True)


def lloyds():
    """Lloyd with the per-wave candidate filter (LDS / DPP kernel; from 2^18 samples on the sorted order, the packed-fp32 screening and
    the distance bounds -- `lloyds_list` as the family name forces their list form from the first sorted iteration) == Lloyd with the
    full scan == the filter in the caller's order without bounds (an experiments build: the full scan in the scalar-record kernel and
    the mask form without screening instead): identical labels, iteration counts, centres."""
    n, ties, t0 = 0, 0, time.time()
    ctxs = []
    exp = bool(_lib.load_library().brov_experiments_build())
    for v in ((0, 257, 16 + 128) if exp else (0, 1, 2 + 4)):
        c = _lib.Context(0)
        c.set_kmeans_variant(v)
        if LIST_FORM:
            c.set_kmeans_bounds_rate(1.0)
        ctxs.append(c)
    while time.time() - t0 < budget:
        N = int(rng.choice([70, 1000, 4097, 30000, 120000, 270000]))
        n_ = int(rng.choice([3, 5, 12, 13, 14, 15]))
        k = int(rng.choice([2, 63, 64, 65, 128, 300, 512, 700, 1024]))
        if k > N:
            continue
        X = np.cumsum(rng.normal(0, 0.05, (N, n_)), 0) * float(rng.choice([1e-3, 1.0, 50.0]))
        if rng.random() < 0.3:
            X = X[rng.permutation(N)]
        rounded = rng.random() < 0.2
        if rounded:
            X = np.round(X, 1)                        # many exact duplicates / ties
        C0 = X[rng.choice(N, k, replace=False)].copy()
        mean = X.mean(0)
        it = int(rng.choice([1, 3, 12]) if N < 200000 else rng.choice([12, 50]))
        (Ca, la, ia, na), (Cb, lb, ib, nbb), (Cc, lc, ic, ncc) = [engine.kmeans_lloyd(X, C0 - mean, max_iter=it, tol_abs=0.0, mean=mean, ctx=c) for c in ctxs]
        assert ncc == nbb and np.array_equal(Cc, Cb) and np.array_equal(lc, lb), ("lloyd mask form", N, n_, k, it)
        assert na == nbb, ("lloyd iterations", N, n_, k, it)
        # round 4: integer member sums -- both variants form the same centres bit for bit, so their labels have nothing to differ by
        # (round 3 had to excuse "exact-distance ties" here: fp64 atomics in arrival order moved the centres between two runs)
        assert np.array_equal(Ca, Cb), ("lloyd centres", N, n_, k, it, float(np.max(np.abs(Ca - Cb))))
        assert np.array_equal(la, lb), ("lloyd labels", N, n_, k, it, int(np.sum(la != lb)))
        n += 1
    for c in ctxs:
        c.close()
    print(f"Lloyd      : {n} cases, candidate filter == full scan (labels, iterations, centres bit for bit), {ties} excused", flush=True)


def copies():
    """Round 6: host <-> device copies of the C ABI (staged through the ctx's pinned blocks between 64 KB and 16 MB, the runtime's own paths
    outside; downloads above one block stream through both) and the pooled brov_malloc -- random sizes around every boundary, odd byte
    counts, several live blocks, round trips bit for bit; DevArray views."""
    import ctypes
    ctx = _lib.default_context(0)
    lib = ctx.lib
    edges = [1, 7, 8, 4096, (64 << 10) - 8, 64 << 10, (64 << 10) + 8, (4 << 20) - 16, 4 << 20, (4 << 20) + 24, (16 << 20) - 8, 16 << 20, (16 << 20) + 8,
             (32 << 20) - 8, 32 << 20, (32 << 20) + 8, (64 << 20) + 40, 70_000_001]
    n, t0 = 0, time.time()
    live = []
    while time.time() - t0 < budget:
        nbytes = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, 40 << 20))
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        p_ = ctypes.c_void_p()
        ctx.check(lib.brov_malloc(ctx.h, nbytes, ctypes.byref(p_)), "brov_malloc")
        ctx.check(lib.brov_memcpy_h2d(ctx.h, p_, src.ctypes.data, nbytes), "h2d")
        live.append((p_, src))
        if len(live) > 6 or rng.random() < 0.5:
            q_, want = live.pop(int(rng.integers(len(live))))
            out = np.empty_like(want)
            ctx.check(lib.brov_memcpy_d2h(ctx.h, out.ctypes.data, q_, want.nbytes), "d2h")
            assert np.array_equal(out, want), ("copy round trip", want.nbytes)
            ctx.check(lib.brov_free(ctx.h, q_), "brov_free")
        n += 1
    for q_, want in live:
        out = np.empty_like(want)
        ctx.check(lib.brov_memcpy_d2h(ctx.h, out.ctypes.data, q_, want.nbytes), "d2h")
        assert np.array_equal(out, want), ("copy round trip", want.nbytes)
        ctx.check(lib.brov_free(ctx.h, q_), "brov_free")
    a = rng.normal(size=(int(rng.integers(2, 3000)), 13))
    d = engine.DevArray.from_host(ctx, a)
    i, j = sorted(int(v) for v in rng.integers(0, len(a) + 1, 2))
    assert np.array_equal(d.rows(i, j).numpy(), a[i:j]) and np.array_equal(d.view(-1).numpy(), a.ravel())
    print(f"copies     : {n} cases (1 B .. 70 MB), every round trip bit for bit", flush=True)


def linear_multistep():
    """Round 6: multistep_se_linear (one pass, explicit powers of A) == multistep_se (H propagated steps) on random shapes."""
    worst, n, t0 = 0.0, 0, time.time()
    while time.time() - t0 < budget:
        n_ = int(rng.choice([3, 12, 13, 16]))
        r = int(rng.choice([1, 6, 8]))
        k = int(rng.choice([1, 17, 48, 200, 512]))
        N = int(rng.choice([2, 50, 257, 3000]))
        H = int(rng.choice([0, 1, 2, 9, 40, 130]))
        X = np.cumsum(rng.normal(0, 0.05, (N, n_)), 0)
        U = rng.uniform(-1, 1, (N, r))
        C = X[rng.choice(N, k, replace=True)] + rng.normal(0, 0.01, (k, n_))
        d = n_ + k
        A = rng.normal(0, 0.6 / np.sqrt(d), (d, d))               # spectral radius ~0.6: the powers stay tame
        B = rng.normal(0, 0.3, (d, r))
        g = float(rng.choice([0.3, 1.0, 3.0]))
        sa, xa = engine.multistep_se(X, U, C, g, A, B, H, want_xhat=True)
        sb, xb = engine.multistep_se_linear(X, U, C, g, A, B, H, want_xhat=True)
        if N - H > 0:
            e = float(np.max(np.abs(xa - xb)) / max(1.0, np.max(np.abs(xa))))
            assert e < 1e-10 and abs(sa - sb) <= 1e-9 * max(1.0, sa), ("linear multistep", n_, r, k, N, H, e)
            worst = max(worst, e)
        else:
            assert sa == sb == 0.0
        n += 1
    print(f"linear mstp: {n} cases, worst rel err {worst:.2e}", flush=True)


LIST_FORM = False


def lloyds_list():
    global LIST_FORM
    LIST_FORM = True
    try:
        lloyds()
    finally:
        LIST_FORM = False


if __name__ == "__main__":
    fams = dict(copies=copies, linear_multistep=linear_multistep, applies=applies, lloyds=lloyds, lloyds_list=lloyds_list, rollouts=rollouts, windows=windows, grams=grams, multistep=multistep, kmeanspp=kmeanspp)
    for name in (sys.argv[3].split(",") if len(sys.argv) > 3 else list(fams)):
        fams[name]()
    print("stress parity: ok")
