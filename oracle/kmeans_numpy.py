"""NumPy restatement of scikit-learn 1.7.2's Lloyd loop.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

The reference gets its RBF centres from `KMeans(n_clusters, n_init="auto", random_state=0).fit(X)`
(Koopman/koopmanEDMDc.py:85,126; scikit-learn 1.7.2, uv.lock:1338-1339) -- a third-party dependency whose
algorithm is restated here from its published source so that the product's device loop (csrc/kmeans.hip) can
be checked step by step, including the parts a run rarely reaches:

  sklearn/cluster/_kmeans.py       `_kmeans_single_lloyd`  (loop, strict convergence, tolerance, final E-step)
  sklearn/cluster/_k_means_lloyd.pyx  `lloyd_iter_chunked_dense` / `_update_chunk_dense`  (E-step: argmin of
                                   |c|^2 - 2 x.c, first minimum; M-step sums)
  sklearn/cluster/_k_means_common.pyx `_relocate_empty_clusters_dense` (empty clusters move to the farthest samples),
                                   `_average_centers` (in-place loop: a still-empty cluster in FRONT of the biggest one
                                   copies its un-averaged SUM, one behind it its mean), `_center_shift`

Pinned by tests/test_oracle_golden.py against scikit-learn itself (importable here and on the GPU box) on sets that
force empty clusters, and by tests/golden/kmeans_empty.npz.  Everything works on data the caller has centred already
(scikit-learn subtracts the column means first and adds them back to the centres at the end).
"""
import numpy as np


def far_rows_numpy(distances, n_empty):
    """`np.argpartition(distances, -n_empty)[:-n_empty-1:-1]` -- scikit-learn's own expression."""
    return np.argpartition(distances, -n_empty)[:-n_empty - 1:-1]


def far_rows_sorted(distances, n_empty):
    """The library's fallback rule (csrc/capi.hip: far_select_default): descending distance, equal ones by ascending row."""
    order = np.lexsort((np.arange(len(distances)), -distances))
    return order[:n_empty]


def e_step(X, C):
    """labels = first argmin over c of (|c|^2 - 2 x.c); sklearn forms it with a GEMM per chunk of 256 samples."""
    c2 = np.einsum("ij,ij->i", C, C)
    labels = np.empty(len(X), dtype=np.int32)
    for s in range(0, len(X), 65536):
        D = c2[None, :] - 2.0 * (X[s:s + 65536] @ C.T)
        labels[s:s + 65536] = np.argmin(D, axis=1)
    return labels


def m_step(X, C_old, labels, far_rows=far_rows_numpy):
    """Sums, relocation of empty clusters, averaging.  Returns (C_new, relocated: bool)."""
    k, n = C_old.shape
    sums = np.zeros((k, n))
    np.add.at(sums, labels, X)
    w = np.bincount(labels, minlength=k).astype(float)
    relocated = False
    empty = np.where(w == 0)[0]
    if len(empty):
        distances = ((X - C_old[labels]) ** 2).sum(axis=1)
        if np.max(distances) != 0:
            far = far_rows(distances, len(empty))
            for idx, new_c in enumerate(empty):
                fi = int(far[idx])
                old_c = labels[fi]
                sums[old_c] -= X[fi]
                sums[new_c] = X[fi]
                w[new_c] = 1.0
                w[old_c] -= 1.0
            relocated = True
    # `_average_centers`: in place, in index order
    arg = int(np.argmax(w))
    C_new = sums
    for j in range(k):
        if w[j] > 0:
            C_new[j] *= 1.0 / w[j]
        else:
            C_new[j] = C_new[arg]            # the SUM row of the biggest cluster when j < arg
    return C_new, relocated


def lloyd(X, C0, max_iter=300, tol_abs=0.0, far_rows=far_rows_numpy):
    """`_kmeans_single_lloyd` on centred data: returns (centres, labels, inertia, n_iter, relocations)."""
    X = np.ascontiguousarray(X, dtype=float)
    C = np.array(C0, dtype=float)
    labels_old = np.full(len(X), -1, dtype=np.int32)
    strict = False
    n_iter, nreloc = 0, 0
    labels = labels_old
    for it in range(max_iter):
        labels = e_step(X, C)
        C_new, rel = m_step(X, C, labels, far_rows)
        nreloc += int(rel)
        shift = ((C_new - C) ** 2).sum()
        C = C_new
        n_iter = it + 1
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift <= tol_abs:
            break
        labels_old = labels
    if not strict:
        labels = e_step(X, C)
    inertia = float(((X - C[labels]) ** 2).sum())
    return C, labels, inertia, n_iter, nreloc
