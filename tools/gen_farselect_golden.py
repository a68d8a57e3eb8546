"""tests/golden/farselect.npz: np.argpartition(distances, -n_empty)[:-n_empty-1:-1] -- the rows scikit-learn's
`_relocate_empty_clusters_dense` moves its empty clusters to (sklearn/cluster/_k_means_common.pyx; the reference reaches it through
KMeans(...).fit, Koopman/koopmanEDMDc.py:85,126) -- from NumPy ITSELF with its SIMD dispatch disabled, i.e. from NumPy's own
introselect (numpy/_core/src/npysort/selection.cpp), which is what the library's rule restates (csrc/capi.hip: npysel).  On x86 hosts
with AVX-512 / AVX2 NumPy dispatches np.argpartition to x86-simd-sort instead and returns the n_empty largest in another order; the
script records that host's answer too (key `*_simd`, informational: it shows that the reference's own result depends on the host).

    python tools/gen_farselect_golden.py            (re-runs itself in a child with NPY_DISABLE_CPU_FEATURES set)
"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DISABLE = "AVX512F AVX512CD AVX512_KNL AVX512_KNM AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"


def cases():
    rng = np.random.default_rng(20251005)
    out = []
    for N, ne in ((2, 1), (3, 1), (3, 2), (5, 2), (50, 3), (64, 1), (1000, 3), (1000, 7), (1000, 40), (4097, 1), (4097, 4), (30000, 2),
                  (30000, 5), (30000, 64)):
        out.append((f"uniform_{N}_{ne}", rng.random(N), ne))
    out.append(("ties_5000_4", np.round(rng.random(5000), 2), 4))                        # many equal distances
    out.append(("ties_777_9", np.round(rng.random(777), 1), 9))
    d = rng.random(3000); d[[5, 17, 2999]] = np.nan
    out.append(("nan_3000_5", d, 5))                                                      # NaNs sort last (= farthest)
    out.append(("sorted_6000_6", np.sort(rng.random(6000)), 6))
    out.append(("reversed_6000_6", np.sort(rng.random(6000))[::-1].copy(), 6))
    out.append(("constant_600_3", np.full(600, 0.25), 3))
    # the shape the depth limit is there for: an adversarial-ish organ pipe, long enough that median-of-3 degrades
    op = np.concatenate([np.arange(8000.0), np.arange(8000.0)[::-1]])
    out.append(("organ_pipe_16000_5", op, 5))
    # Musser's median-of-3 killer: the partitions stop making progress and the depth limit hands over to the median of medians of 5
    # (reached here with a small kth, i.e. a large n_empty: the code path is the same for any kth)
    for n_k in (2000, 8000):
        kk = n_k // 2
        a = [0] * n_k
        for i in range(1, kk + 1):
            if i % 2 == 1:
                a[i - 1], a[i] = i, kk + i
            a[kk + i - 1] = 2 * i
        out.append((f"killer_{n_k}_{n_k - 5}", np.array(a, dtype=float), n_k - 5))
    # squared distances of a k-means run with duplicate centres: the real thing (clustered values, exact ties at 0)
    X = rng.normal(0, 1, (6000, 12)); C = X[rng.choice(6000, 20, replace=False)]
    lab = np.argmin(((X[:, None, :] - C[None]) ** 2).sum(-1), axis=1)
    out.append(("kmeans_6000_3", ((X - C[lab]) ** 2).sum(axis=1), 3))
    return out


def run():
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    res = {}
    for name, d, ne in cases():
        res[name + "_d"] = d
        res[name + "_n"] = np.int64(ne)
        res[name + "_far"] = np.argpartition(d, -ne)[:-ne - 1:-1].astype(np.int64)
    res["simd_dispatch_enabled"] = np.array(sorted(k for k, v in feat.items() if v and k.startswith(("AVX512", "AVX2"))))
    return res


if __name__ == "__main__":
    if os.environ.get("FARSELECT_CHILD") == "1":
        res = run()
        # np.argpartition's SIMD kernels (x86-simd-sort) are dispatched for AVX512_SKX and AVX2: both must be off in this child
        assert not {"AVX512_SKX", "AVX512F", "AVX2"} & set(res["simd_dispatch_enabled"].tolist()), res["simd_dispatch_enabled"]
        np.savez_compressed(sys.argv[1], **res)
    else:
        tmp = os.path.join(REPO, "tests", "golden", "_farselect_generic.npz")
        env = dict(os.environ, FARSELECT_CHILD="1", NPY_DISABLE_CPU_FEATURES=DISABLE)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tmp], env=env)
        gen = dict(np.load(tmp))
        os.remove(tmp)
        here = run()                                                  # this host's dispatch (informational)
        differ = []
        for k in list(gen):
            if k.endswith("_far"):
                gen[k[:-4] + "_far_simd"] = here[k]
                if not np.array_equal(here[k], gen[k]):
                    differ.append(k[:-4])
        gen["numpy_version"] = np.array(np.__version__)
        gen["simd_host_features"] = here["simd_dispatch_enabled"]
        del gen["simd_dispatch_enabled"]
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "farselect.npz"), **gen)
        print("wrote tests/golden/farselect.npz;", len(differ), "cases where this host's SIMD path answers differently:", differ)
