"""Hunt for a Lloyd filter / full scan label mismatch (stress family `lloyds`) and print what kind it is."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctxs = []
for v in (0, 1):
    c = _lib.Context(0); c.set_kmeans_variant(v); ctxs.append(c)
found = 0
for trial in range(3000):
    N = int(rng.choice([4097, 1000, 30000])); n_ = int(rng.choice([15, 12, 13, 5])); k = int(rng.choice([700, 512, 300, 65]))
    scale = float(rng.choice([1e-3, 1.0, 50.0]))
    X = np.cumsum(rng.normal(0, 0.05, (N, n_)), 0) * scale
    shuf = rng.random() < 0.3; rnd = rng.random() < 0.2
    if shuf: X = X[rng.permutation(N)]
    if rnd: X = np.round(X, 1)
    C0 = X[rng.choice(N, k, replace=False)].copy()
    mean = X.mean(0)
    it = int(rng.choice([1, 2, 3]))
    (Ca, la, ia, na), (Cb, lb, ib, nb) = [engine.kmeans_lloyd(X, C0 - mean, max_iter=it, tol_abs=0.0, mean=mean, ctx=c) for c in ctxs]
    bad = np.nonzero(la != lb)[0]
    if len(bad):
        found += 1
        i = int(bad[0])
        # the labels belong to the last E-step, whose centres are the returned ones
        x = (X[i] - mean).astype(np.longdouble)
        d = ((x[None, :] - Cb.astype(np.longdouble)) ** 2).sum(1)
        order = np.argsort(d)
        print(f"trial {trial}: N={N} n={n_} k={k} scale={scale} shuffled={shuf} rounded={rnd} it={it}: {len(bad)} labels differ; sample {i}: filter -> {la[i]}, full -> {lb[i]}")
        print("   d2[filter label] =", float(d[la[i]]), " d2[full label] =", float(d[lb[i]]), " rel diff =", float((d[la[i]] - d[lb[i]]) / max(d[lb[i]], 1e-300)),
              " nearest three:", [(int(j), float(d[j])) for j in order[:3]])
        print("   centres equal (dup)?", bool(np.array_equal(Cb[la[i]], Cb[lb[i]])), " |x|^2 =", float((x * x).sum()), " max|c|^2 =", float((Cb ** 2).sum(1).max()),
              " centre diff filter-vs-full runs:", float(np.abs(Ca - Cb).max()))
        if found >= 5: break
print("mismatching cases:", found)
