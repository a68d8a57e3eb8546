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


def _less(a, b):
    """NumPy's order for floating point: NaNs are the largest (npy::double_tag::less)."""
    return a < b or (b != b and a == a)


class _Introselect:
    """NumPy's OWN selection algorithm on an index array (numpy/_core/src/npysort/selection.cpp, `introselect_<Tag, arg=true>`):
    median-of-3 quickselect, median of medians of 5 once 2 floor(log2 n) partitions are spent, an O(n kth) selection when kth is within 3
    of the range's start, a maximum scan for kth = n - 1.  This is what np.argpartition runs where no SIMD kernel is dispatched, and
    the rule the library restates (csrc/capi.hip: npysel); pinned by tests/golden/farselect.npz (NumPy itself with its dispatch off)."""

    def __init__(self, v):
        self.v = [float(x) for x in v]
        self.t = list(range(len(v)))
        self.used_median_of_medians = False

    def at(self, i):
        return self.v[self.t[i]]

    def swap(self, i, j):
        self.t[i], self.t[j] = self.t[j], self.t[i]

    def dumb_select(self, base, num, kth):
        for i in range(kth + 1):
            minidx, minval = i, self.at(base + i)
            for k in range(i + 1, num):
                if _less(self.at(base + k), minval):
                    minidx, minval = k, self.at(base + k)
            self.swap(base + i, base + minidx)

    def median3_swap(self, low, mid, high):
        if _less(self.at(high), self.at(mid)):
            self.swap(high, mid)
        if _less(self.at(high), self.at(low)):
            self.swap(high, low)
        if _less(self.at(low), self.at(mid)):
            self.swap(low, mid)
        self.swap(mid, low + 1)

    def median5(self, b):
        at, swap = self.at, self.swap
        if _less(at(b + 1), at(b)):
            swap(b + 1, b)
        if _less(at(b + 4), at(b + 3)):
            swap(b + 4, b + 3)
        if _less(at(b + 3), at(b)):
            swap(b + 3, b)
        if _less(at(b + 4), at(b + 1)):
            swap(b + 4, b + 1)
        if _less(at(b + 2), at(b + 1)):
            swap(b + 2, b + 1)
        if _less(at(b + 3), at(b + 2)):
            return 1 if _less(at(b + 3), at(b + 1)) else 3
        return 2

    def median_of_median5(self, base, num):
        self.used_median_of_medians = True
        nmed = num // 5
        for i in range(nmed):
            m = self.median5(base + 5 * i)
            self.swap(base + 5 * i + m, base + i)
        if nmed > 2:
            self.select(base, nmed, nmed // 2)
        return nmed // 2

    def select(self, base, num, kth):
        low, high = 0, num - 1
        if kth - low < 3:
            self.dumb_select(base + low, high - low + 1, kth - low)
            return
        if kth == num - 1:
            maxidx, maxval = low, self.at(base + low)
            for k in range(low + 1, num):
                if not _less(self.at(base + k), maxval):
                    maxidx, maxval = k, self.at(base + k)
            self.swap(base + kth, base + maxidx)
            return
        depth_limit = 2 * (int(num).bit_length() - 1)
        while low + 1 < high:
            ll, hh = low + 1, high
            if depth_limit > 0 or hh - ll < 5:
                self.median3_swap(base + low, base + low + (high - low) // 2, base + high)
            else:
                mid = ll + self.median_of_median5(base + ll, hh - ll)
                self.swap(base + mid, base + low)
                ll -= 1
                hh += 1
            depth_limit -= 1
            pivot = self.at(base + low)
            while True:                                   # unguarded_partition_
                ll += 1
                while _less(self.at(base + ll), pivot):
                    ll += 1
                hh -= 1
                while _less(pivot, self.at(base + hh)):
                    hh -= 1
                if hh < ll:
                    break
                self.swap(base + ll, base + hh)
            self.swap(base + low, base + hh)
            if hh >= kth:
                high = hh - 1
            if hh <= kth:
                low = ll
        if high == low + 1 and _less(self.at(base + high), self.at(base + low)):
            self.swap(base + high, base + low)


def far_rows_introselect(distances, n_empty):
    """`np.argpartition(distances, -n_empty)[:-n_empty-1:-1]` as NumPy's own (non-SIMD) introselect returns it: the library's rule
    (csrc/capi.hip: far_select_default) for callers without the NumPy callback and for sharded runs."""
    N = len(distances)
    s = _Introselect(distances)
    s.select(0, N, N - n_empty)
    return np.array(s.t[::-1][:n_empty], dtype=np.int64)


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


# ---- the device loop's own arithmetic (csrc/kmeans.hip, round 4): member sums in 2^-48 fixed point ------------------------------
# A stand-in for the HIP loop where there is no GPU (the world-size-2 gloo test of the sharded Lloyd) and a bit-for-bit checker
# of its centres where there is one: the same quantisation (x -> round-half-even(x * 2^(48 - e_j)), 2^e_j > max |x_j|), integer
# sums, the same conversion back, the same empty-cluster rules.  Only the E-step differs in rounding (a GEMM here, an FMA chain
# there): labels can differ for a sample within ~1e-16 of a tie, nothing else can.
FIX_BITS = 48
LIMB = 42


def fix_scales(colmax):
    """s_j = 2^(48 - e_j) with frexp's e_j (2^e_j > max_j); max_j = 0 -> e_j = 0.  Returns (s, 1/s)."""
    e = np.where(colmax > 0, np.frexp(np.where(colmax > 0, colmax, 1.0))[1], 0)
    sh = np.clip(FIX_BITS - e, -1000, 1000)
    return np.ldexp(1.0, sh), np.ldexp(1.0, -sh)


def _to_double(t):
    """kmeans.hip: km_to_double (sign and magnitude; one rounding below 2^64, two above)."""
    m = abs(int(t))
    d = float(m >> 64) * 18446744073709551616.0 + float(m & ((1 << 64) - 1))
    return -d if t < 0 else d


def _int_sums(Q, labels, k):
    """exact per-cluster sums of the int64 rows Q as Python integers [k][n] (two 24-bit-split passes keep np.add.at inside int64)"""
    hi, lo = Q >> 24, Q & ((1 << 24) - 1)
    sh = np.zeros((k, Q.shape[1]), dtype=np.int64)
    sl = np.zeros((k, Q.shape[1]), dtype=np.int64)
    np.add.at(sh, labels, hi)
    np.add.at(sl, labels, lo)
    return [[(int(sh[c, j]) << 24) + int(sl[c, j]) for j in range(Q.shape[1])] for c in range(k)]


def to_limbs(tot, cnt, changed):
    """[k][n+1][2] int64 limbs (hi * 2^42 + lo, 0 <= lo < 2^42) + the 2-word tail, the layout the ranks all-reduce (kmeans.hip: red)"""
    k, n = len(tot), len(tot[0])
    out = np.zeros(k * (n + 1) * 2 + 2, dtype=np.int64)
    for c in range(k):
        for j in range(n + 1):
            v = tot[c][j] if j < n else int(cnt[c])
            out[(c * (n + 1) + j) * 2] = v >> LIMB
            out[(c * (n + 1) + j) * 2 + 1] = v & ((1 << LIMB) - 1)
    out[-2] = changed
    return out


def from_limbs(buf, k, n):
    tot = [[(int(buf[(c * (n + 1) + j) * 2]) << LIMB) + int(buf[(c * (n + 1) + j) * 2 + 1]) for j in range(n)] for c in range(k)]
    cnt = [(int(buf[(c * (n + 1) + n) * 2]) << LIMB) + int(buf[(c * (n + 1) + n) * 2 + 1]) for c in range(k)]
    return tot, cnt, int(buf[-2])


def lloyd_fixed_point(X, C0, max_iter=300, tol_abs=0.0, allreduce=None, far_rows=far_rows_introselect, shard=None):
    """The device loop (edmdc_kmeans_lloyd_dev) restated: X = this rank's centred rows, C0 the common initial centres.
    allreduce(int64 array, op) combines a buffer over the ranks in place (op 0: sum, op 1: max) -- None: one rank.
    shard = (global index of this rank's first row, rows over all ranks) for a sharded run.
    Returns (centres, labels, n_iter, relocations).  Empty clusters, sharded: the ranks' distances are put side by side in global row
    order (a sum in which every rank fills its own rows), every rank applies `far_rows` to that array, and the owner of each chosen row
    sends its label and fixed-point coordinates -- as csrc/capi.hip: kmeans_relocate does."""
    X = np.ascontiguousarray(X, dtype=float)
    N, n = X.shape
    k = len(C0)
    finite = np.isfinite(X)
    colmax = np.where(finite, np.abs(X), 0.0).max(axis=0) if N else np.zeros(n)
    rng = np.zeros(16, dtype=np.int64)
    rng[:n] = colmax.view(np.int64)                     # bit patterns of non-negative doubles order like integers
    if allreduce is not None:
        allreduce(rng, 1)
    s, inv_s = fix_scales(rng[:n].view(np.float64))
    Q = np.rint(X * s).astype(np.int64)
    C = np.array(C0, dtype=float)
    labels_old = np.full(N, -1, dtype=np.int32)
    n_iter, nreloc = 0, 0
    for it in range(max_iter):
        labels = e_step(X, C)
        tot = _int_sums(Q, labels, k)
        cnt = np.bincount(labels, minlength=k)
        buf = to_limbs(tot, cnt, int(np.sum(labels != labels_old)))
        if allreduce is not None:
            allreduce(buf, 0)
        tot, cnt, changed = from_limbs(buf, k, n)
        empty = [c for c in range(k) if cnt[c] == 0]
        if empty:
            local = ((X - C[labels]) ** 2).sum(axis=1)
            row0 = 0
            if allreduce is not None:
                row0, Ng = shard
                g = np.zeros(Ng, dtype=np.int64)
                g[row0:row0 + N] = local.view(np.int64)
                allreduce(g, 0)
                distances = g.view(np.float64)
            else:
                distances = local
            if np.max(distances) != 0:
                far = far_rows(distances, len(empty))
                for idx, new_c in enumerate(empty):
                    fi = int(far[idx]) - row0
                    msg = np.zeros(18, dtype=np.int64)
                    if 0 <= fi < N:
                        msg[:n] = Q[fi]
                        msg[16], msg[17] = labels[fi], 1
                    if allreduce is not None:
                        allreduce(msg, 0)
                    assert msg[17] == 1, "exactly one rank owns a relocated row"
                    old_c = int(msg[16])
                    for j in range(n):
                        tot[old_c][j] -= int(msg[j])
                        tot[new_c][j] = int(msg[j])
                    cnt[new_c] = 1
                    cnt[old_c] -= 1
                nreloc += 1
        arg = int(np.argmax(cnt))
        C_new = np.empty_like(C)
        for c in range(k):
            for j in range(n):
                if cnt[c] > 0:
                    C_new[c, j] = (_to_double(tot[c][j]) * inv_s[j]) / float(cnt[c])
                elif c > arg:
                    C_new[c, j] = (_to_double(tot[arg][j]) * inv_s[j]) / float(cnt[arg])
                else:
                    C_new[c, j] = _to_double(tot[arg][j]) * inv_s[j]
        shift = ((C_new - C) ** 2).sum()
        C = C_new
        n_iter = it + 1
        if changed == 0:
            break
        if shift <= tol_abs:
            break
        labels_old = labels
    return C, e_step(X, C), n_iter, nreloc


# ---- distance bounds of the sorted loop (csrc/kmeans.hip: kmeans_bounds_kernel, round 4), restated for a property test ----------------
BND_TOP = 4          # movers taken apart (kmeans.hip: KM_BND_TOP)


def bounds_step(ub, lb, labels, C_old, C_new):
    """One M-step's effect on Hamerly's bounds, as the device applies it: ub >= d(x, c_a) grows by the own centre's shift; lb <= d(x, c)
    for every other centre gives way by the largest shift among the centres that are not among the BND_TOP largest movers, and for
    each of those movers t != a by its own shift -- unless the mover is far from the sample's own centre NOW:
        lb' = min(lb - m_rest, min_t max(lb - shift_t, d(c_a', c_t') - ub')).
    Returns (ub', lb').  (Exact arithmetic here; the device rounds every step away from "skip".)"""
    shift = np.sqrt(((C_new - C_old) ** 2).sum(axis=1))
    order = np.argsort(-shift, kind="stable")
    movers = order[:BND_TOP]
    m_rest = shift[order[BND_TOP]] if len(shift) > BND_TOP else 0.0
    ubn = ub + shift[labels]
    lbn = lb - m_rest
    for t in movers:
        dm = np.sqrt(((C_new[labels] - C_new[t]) ** 2).sum(axis=1))
        cand = np.maximum(lb - shift[t], dm - ubn)
        lbn = np.where(labels == t, lbn, np.minimum(lbn, cand))
    return ubn, lbn
