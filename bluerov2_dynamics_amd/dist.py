"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards naturally:
  * rollouts  -- trajectories are independent: rank i owns the contiguous shard
                 [shard_range(B, i, world)) and never talks to the others;
  * EDMDc fit -- samples are independent: every rank builds its local G^T[G|Y] blocks on its GPU,
                 ONE all-reduce (sum) of a single (p*p + p*d)-double buffer (4.5 MB at k = 512)
                 combines them, then every rank solves the same p x p system on the host.
No other exchange exists; centres are broadcast once.
"""
import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous, balanced split of range(total): returns (start, stop) of `rank`."""
    assert 0 <= rank < world
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def world_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_gram_(GtG, GtY, group=None, deterministic=False):
    """In-place sum over ranks of both blocks with a single collective.  GtG/GtY: torch tensors
    (CUDA for nccl, CPU for gloo).  Returns (GtG, GtY).
    deterministic=True: all-gather the per-rank blocks and add them in rank order on every rank (world x 4.5 MB at
    k = 512) -- the same bits for a given sharding whatever algorithm the collective library picks; the all-reduce
    leaves the order of the additions to RCCL (ring / tree, chunked)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return GtG, GtY
    p2, pd = GtG.numel(), GtY.numel()
    buf = torch.empty(p2 + pd, dtype=GtG.dtype, device=GtG.device)
    buf[:p2].copy_(GtG.reshape(-1))
    buf[p2:].copy_(GtY.reshape(-1))
    if deterministic:
        world = dist.get_world_size(group)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)          # supported by both nccl (RCCL) and gloo
        buf = parts[0].clone()
        for w in range(1, world):
            buf += parts[w]
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    GtG.copy_(buf[:p2].view_as(GtG))
    GtY.copy_(buf[p2:].view_as(GtY))
    return GtG, GtY


def broadcast_centers_(C, src=0, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(C, src, group=group)
    return C


def _host(t):
    """host copy of a device array of either kind (engine.DevArray or torch tensor)"""
    from . import engine
    return t.numpy() if isinstance(t, engine.DevArray) else t.cpu().numpy()


def _zeros(X, shape):
    """zero-filled fp64 device array of X's kind (engine.DevArray or torch CUDA tensor)"""
    from . import engine
    if isinstance(X, engine.DevArray):
        return engine.DevArray(X.ctx, shape).zero_()
    import torch
    return torch.zeros(shape, dtype=torch.float64, device=X.device)


def _device_gram(X, U, C, gamma, nbags, L, xs, us):
    """Local Gram on this rank's GPU (X, U, C: fp64 device arrays -- torch CUDA tensors or engine.DevArray) -> (GtG, GtY) of the same kind."""
    from . import engine
    n, r, k = X.shape[-1], U.shape[-1], C.shape[0]
    p, d = n + k + r, n + k
    GtG, GtY = _zeros(X, (p, p)), _zeros(X, (p, d))
    engine.gram_dev(X.reshape(-1, n), U.reshape(-1, r), C, gamma, nbags, L, xs, us, GtG, GtY)
    return GtG, GtY


def _device_gtg(X, U, C, gamma, nbags, L, xs, us):
    """Local G^T G alone (fit()'s own order never forms G^T Y) -> (GtG, a one-element placeholder)."""
    from . import engine
    n, r, k = X.shape[-1], U.shape[-1], C.shape[0]
    p = n + k + r
    GtG = _zeros(X, (p, p))
    engine.gram_dev(X.reshape(-1, n), U.reshape(-1, r), C, gamma, nbags, L, xs, us, GtG, None)
    return GtG, _zeros(X, (1,))


def _device_apply(X, U, C, gamma, nbags, L, xs, us, P):
    """Local (P G^T) Y on this rank's GPU -> M [p, d] device array."""
    from . import engine
    n, r, k = X.shape[-1], U.shape[-1], C.shape[0]
    M = _zeros(X, (n + k + r, n + k))
    engine.pinv_apply_dev(X.reshape(-1, n), U.reshape(-1, r), C, gamma, nbags, L, xs, us, P, M)
    return M


def fit_sharded(X_local, U_local, C, gamma, ridge, gram_fn=None, group=None, deterministic=False, order="fit_multi", apply_fn=None,
                allreduce=None, pinv="auto"):
    """EDMDc fit over trajectories sharded across ranks.

    X_local [nb_local, L+1, n], U_local [nb_local, L, r]: this rank's bags (torch tensors, or engine.DevArray together with
    allreduce=_lib.Comm(...).allreduce_gram_ -- a sharded fit with no torch in the process);
    C [k, n] centres (identical on all ranks -- broadcast them first).
    gram_fn(X, U, C, gamma, nbags, L, xs, us) -> (GtG, GtY) defaults to the HIP path; tests on CPU
    inject a host implementation to exercise the collective + solve plumbing under gloo.
    order="fit_multi": M = P (G^T Y) (Koopman/koopmanEDMDc.py:147), ONE collective.
    order="fit"      : M = (P G^T) Y (:97): P from the all-reduced Gram is identical on every rank, every rank forms its
                       local sum over its own pairs (apply_fn, default the HIP path) and a second all-reduce of the p x d
                       block adds them up.
    allreduce(a, b)  : in-place sum over ranks of two tensors (default: torch.distributed through allreduce_gram_; the
                       torch-free communicator of the C ABI passes _lib.Comm.allreduce_gram_).
    Returns (A [d,d], B [d,r]) as NumPy arrays, identical on every rank."""
    import numpy as np
    from . import engine
    nb, L1, n = X_local.shape
    L = L1 - 1
    assert U_local.shape[0] == nb and U_local.shape[1] == L
    gram_fn = gram_fn or (_device_gtg if order == "fit" else _device_gram)
    allreduce = allreduce or (lambda a, b: allreduce_gram_(a, b, group, deterministic=deterministic))
    GtG, GtY = gram_fn(X_local, U_local, C, gamma, nb, L, L + 1, L)
    allreduce(GtG, GtY)
    d = n + C.shape[0]
    if order == "fit_multi":
        return engine.solve_AB(_host(GtG), _host(GtY), ridge, d, pinv=pinv)
    if order != "fit":
        raise ValueError("order must be 'fit' or 'fit_multi'")
    with engine._blas_threads():
        P = engine._host_pinv(_host(GtG), ridge, pinv)
    M = (apply_fn or _device_apply)(X_local, U_local, C, gamma, nb, L, L + 1, L, P)
    pad = _zeros(M, (1,))                                  # the collective takes two arrays; the second one is a dummy
    allreduce(M, pad)
    Mt = _host(M).T
    return np.ascontiguousarray(Mt[:, :d]), np.ascontiguousarray(Mt[:, d:])


def rollout_sharded(model, integrator, B_total, T, dt, seed=0x5EED, layout="tub", stride=None, device=None):
    """Every rank rolls out its shard of a B_total-trajectory ensemble of the synthetic control stream
    (values independent of the sharding).  Returns (b0, b1, xT [b1-b0, nx] CUDA tensor, traj or None)."""
    import torch
    from . import engine
    from ._lib import NX, NU
    rank, world = world_info()
    b0, b1 = shard_range(B_total, rank, world)
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    B = b1 - b0
    nu, nx = NU[model], NX[model]
    U = torch.empty((T, nu, B) if layout == "tub" else (B, T, nu), dtype=torch.float64, device=dev)
    engine.fill_controls_dev(U, layout, "iid", seed=seed, b0=b0, T_total=T)
    x0 = torch.zeros((B, nx), dtype=torch.float64, device=dev)
    x0[:, 2] = 5.0
    if model == 2:
        x0[:, 3] = 1.0
    traj = None
    if stride:
        rows = T // stride + 1
        traj = torch.empty((rows, nx, B) if layout == "tub" else (B, rows, nx), dtype=torch.float64, device=dev)
    xT = torch.empty((B, nx), dtype=torch.float64, device=dev)
    engine.rollout_dev(model, integrator, x0, U, dt, traj=traj, xT=xT, layout=layout, stride=stride or 1)
    return b0, b1, xT, traj


# ---- sharded k-means (round 4): the centres of KoopmanEDMDc.fit over ALL ranks' samples (Koopman/koopmanEDMDc.py:85 runs KMeans on all of X) ----
class _DeviceWords:
    """Zero-copy view of `count` int64 words at a raw device address for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i8", "data": (int(ptr), False), "version": 3}


def allreduce_words_(t, op, group=None):
    """In-place all-reduce of an int64 tensor: op 0 = SUM, op 1 = MAX (bit patterns of non-negative doubles order like integers).
    nccl (= RCCL) reduces the device tensor where it lies, ordered behind torch's current stream; any other backend (gloo) goes
    through the host."""
    import torch.distributed as dist
    rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=rop, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=rop, group=group)
    return t


def _shard_place(n_local, device, group=None):
    """(rank, world, global index of this rank's first row, rows over all ranks) for contiguous shards in rank order."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = n_local
    allreduce_words_(counts, 0, group)
    counts = counts.cpu()
    return rank, world, int(counts[:rank].sum().item()), int(counts.sum().item())


def kmeanspp_sharded(X_local, k, mean=None, random_state=0, group=None, ctx=None):
    """k-means++ seeding (scikit-learn's `_kmeans_plusplus`, scikit-learn's random stream) over rows sharded across ranks in rank
    order: every rank passes its rows and ends with the same k centres (centred frame) and the same GLOBAL sample indices -- those
    of the unsharded seeding unless a drawn value falls within rounding of a running-sum boundary.  Two small exchanges per centre
    (csrc/kmeans.hip: launch_kmeanspp_sharded).  One rank (or no process group): the plain device seeding."""
    import torch
    import torch.distributed as dist
    from . import engine
    ctx = ctx or engine.default_context(X_local.device.index)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return engine.kmeanspp_dev(X_local, k, mean=mean, random_state=random_state, ctx=ctx)
    N, n = X_local.shape
    rank, world, row0, Ng = _shard_place(N, X_local.device, group)

    def cb(ptr, count, op):
        allreduce_words_(torch.as_tensor(_DeviceWords(ptr, count), device=X_local.device), op, group)

    ctx.set_kmeans_shard(rank, world, row0, Ng)
    ctx.set_kmeans_allreduce(cb)
    try:
        return engine.kmeanspp_dev(X_local, k, mean=mean, random_state=random_state, ctx=ctx, n_global=Ng)
    finally:
        ctx.set_kmeans_allreduce(None)
        ctx.set_kmeans_shard()


def kmeans_centers_sharded(X_local, k, random_state=0, max_iter=300, tol=1e-4, group=None, ctx=None, timings=None):
    """The RBF centres of KoopmanEDMDc.fit over ALL ranks' rows (Koopman/koopmanEDMDc.py:85: KMeans(k, n_init="auto",
    random_state=0).fit(X) sees every sample): column means and the tolerance's variance over all rows, the k-means++ seeding
    over all rows, Lloyd's loop with one integer all-reduce per iteration.  Every rank returns the same (centres CUDA [k, n],
    inertia over all rows, n_iter); they are the one-rank centres up to the rounding of the column means (the Lloyd loop itself is
    bit-identical for identical means and seeds)."""
    import time
    import torch
    import torch.distributed as dist
    from . import engine
    ctx = ctx or engine.default_context(X_local.device.index)
    mean, var, _ = column_stats_sharded(X_local, group)
    mean_h = mean.cpu().numpy()
    t0 = time.perf_counter()
    C0, _ = kmeanspp_sharded(X_local, k, mean=mean_h, random_state=random_state, group=group, ctx=ctx)
    torch.cuda.synchronize(X_local.device)
    t1 = time.perf_counter()
    C, _, inertia, n_iter = kmeans_lloyd_sharded(X_local, C0, mean=mean_h, max_iter=max_iter, tol_abs=var * tol, group=group, ctx=ctx)
    torch.cuda.synchronize(X_local.device)
    t2 = time.perf_counter()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        it = torch.tensor([inertia], dtype=torch.float64, device=X_local.device)
        if dist.get_backend(group) != "nccl":
            h = it.cpu()
            dist.all_reduce(h, group=group)
            it = h
        else:
            dist.all_reduce(it, group=group)
        inertia = float(it.item())
    if timings is not None:
        timings.update(kmeanspp_s=t1 - t0, lloyd_s=t2 - t1, lloyd_iterations=n_iter)
    return C + mean, inertia, n_iter


def kmeans_lloyd_sharded(X_local, C0, mean=None, max_iter=300, tol_abs=0.0, group=None, lloyd_fn=None, ctx=None, force_exchange=False):
    """Lloyd's loop over rows sharded across ranks: every rank passes its own rows X_local [N_local, n] and the SAME initial
    centres C0 [k, n] (centred frame: C0 and the returned centres are relative to `mean`).  Per iteration one all-reduce (SUM,
    int64) of the member sums -- 2 k (n + 1) + 2 words, 53 KB at k = 512 -- and one all-reduce (MAX) of 16 words before the loop.
    The member sums are integers (csrc/kmeans.hip), so every rank ends with the same centres and they are the centres of the
    unsharded run bit for bit.  Returns (centres [k, n], labels of the local rows, local inertia, n_iter); the inertia of the
    whole set is the sum over ranks.  Empty clusters are relocated to the farthest rows of the WHOLE set: the ranks' distances are
    gathered in global row order and every rank applies the ctx's selection rule to them (np.argpartition through the callback by
    default, like scikit-learn on this host; the library's introselect restatement after Context.set_kmeans_far_select(False)) --
    the rows, and the centres, of the unsharded run under the same rule (include/brov2.h: edmdc_set_kmeans_shard).
    lloyd_fn(X_local, C0, mean, max_iter, tol_abs, allreduce): the loop itself -- default the HIP path
    (edmdc_kmeans_lloyd_dev with edmdc_set_kmeans_allreduce); the CPU tests inject the oracle's stand-in under gloo.
    force_exchange: install the exchange in a process group of ONE rank too (tests: the all-reduce of the library's own device
    words through the group's backend, on a box with a single GPU)."""
    import torch
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange)
    if lloyd_fn is not None:
        ar = (lambda buf, op: allreduce_words_(buf, op, group)) if sharded else None
        return lloyd_fn(X_local, C0, mean, max_iter, tol_abs, ar)
    import ctypes
    from . import engine
    from ._lib import _hptr, as_f64
    ctx = ctx or engine.default_context(X_local.device.index)
    ctx.use_torch_stream()
    N, n = X_local.shape
    k = C0.shape[0]
    C = C0.to(device=X_local.device, dtype=torch.float64).contiguous().clone()
    labels = torch.empty(N, dtype=torch.int32, device=X_local.device)
    inertia, n_iter = ctypes.c_double(0.0), ctypes.c_int(0)
    m = None if mean is None else as_f64(mean).reshape(n)

    def cb(ptr, count, op):
        allreduce_words_(torch.as_tensor(_DeviceWords(ptr, count), device=X_local.device), op, group)

    if sharded:
        # global index of this rank's first row (ranks hold contiguous shards in rank order): the tie-break of an empty cluster's relocation
        rank, world, row0, Ng = _shard_place(N, X_local.device, group)
        ctx.set_kmeans_shard(rank, world, row0, Ng)
        ctx.set_kmeans_allreduce(cb)
    try:
        torch.cuda.current_stream(X_local.device).synchronize()
        ctx.check(ctx.lib.edmdc_kmeans_lloyd_dev(ctx.h, N, n, k, X_local.data_ptr(), X_local.stride(0), _hptr(m), C.data_ptr(), int(max_iter),
                                                 float(tol_abs), labels.data_ptr(), ctypes.byref(inertia), ctypes.byref(n_iter)),
                  "edmdc_kmeans_lloyd_dev (sharded)")
    finally:
        if sharded:
            ctx.set_kmeans_allreduce(None)
            ctx.set_kmeans_shard()
    return C, labels, inertia.value, n_iter.value


def column_stats_sharded(X_local, group=None):
    """(mean [n], mean of the column variances, total rows) over all ranks' rows: per-rank sums all-gathered and added in rank order
    on every rank (the same bits on every rank).  KMeans centres the data with the column means and scales its tolerance with
    the mean variance (sklearn/cluster/_kmeans.py: `_tolerance`)."""
    import torch
    import torch.distributed as dist
    n = X_local.shape[1]
    part = torch.zeros(2 * n + 1, dtype=torch.float64, device=X_local.device)
    part[:n] = X_local.sum(dim=0)
    part[2 * n] = X_local.shape[0]
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def gather(v):
        if world == 1:
            return v
        parts = [torch.empty_like(v) for _ in range(world)]
        if v.is_cuda and dist.get_backend(group) != "nccl":
            hp = [p.cpu() for p in parts]
            dist.all_gather(hp, v.cpu(), group=group)
            parts = [p.to(v.device) for p in hp]
        else:
            dist.all_gather(parts, v, group=group)
        out = parts[0].clone()
        for p in parts[1:]:
            out += p
        return out

    tot = gather(part)
    Nt = tot[2 * n]
    mean = tot[:n] / Nt
    part2 = torch.zeros_like(part)
    part2[:n] = ((X_local - mean) ** 2).sum(dim=0)
    var = gather(part2)[:n] / Nt
    return mean, float(var.mean().item()), int(Nt.item())
