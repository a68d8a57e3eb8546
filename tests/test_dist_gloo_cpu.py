"""world_size-2 `gloo` rehearsal of the multi-GPU EDMDc path on CPU: shard bags over ranks, local
Gram (the oracle stands in for the HIP kernel -- there is no GPU here), ONE all-reduce of the packed
[GtG | GtY] buffer, identical host solve on every rank == single-process fit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_gram(X, U, C, gamma, nbags, L, xs, us):
    from oracle import edmdc_numpy as ek
    Xn, Un, Cn = X.numpy(), U.numpy(), C.numpy()
    GtG, GtY, _ = ek.gram(list(Xn), [np.vstack([u, np.zeros((1, u.shape[1]))]) for u in Un], Cn, gamma)
    return torch.from_numpy(GtG), torch.from_numpy(GtY)


def _oracle_apply(X, U, C, gamma, nbags, L, xs, us, P):
    """(P G^T) Y over this rank's bags, fit()'s own association (Koopman/koopmanEDMDc.py:97), NumPy."""
    from oracle import edmdc_numpy as ek
    Cn = C.numpy()
    M = np.zeros((P.shape[0], Cn.shape[1] + Cn.shape[0]))
    for Xb, Ub in zip(X.numpy(), U.numpy()):
        G = np.hstack([ek.lift(Xb[:-1], Cn, gamma), Ub])
        M += (P @ G.T) @ ek.lift(Xb[1:], Cn, gamma)
    return torch.from_numpy(M)


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:          # surface the failure instead of letting the parent wait for its timeout
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bluerov2_dynamics_amd import dist as bd
    g = load_golden("edmdc.npz")
    X, U = g["X"], g["U"]
    nb, L = 8, 199
    Xb = np.stack([X[i * 200:(i + 1) * 200] for i in range(nb)])
    Ub = np.stack([U[i * 200:i * 200 + L] for i in range(nb)])
    b0, b1 = bd.shard_range(nb, rank, world)
    C = torch.from_numpy(g["centers"].copy()) if rank == 0 else torch.zeros(g["centers"].shape, dtype=torch.float64)
    bd.broadcast_centers_(C)
    A, B = bd.fit_sharded(torch.from_numpy(Xb[b0:b1]), torch.from_numpy(Ub[b0:b1]), C, 1.0, 1e-3, gram_fn=_oracle_gram)
    # fixed-order variant (all-gather + rank-order sum): same reduced system up to the order of two additions
    Ad, Bd = bd.fit_sharded(torch.from_numpy(Xb[b0:b1]), torch.from_numpy(Ub[b0:b1]), C, 1.0, 1e-3, gram_fn=_oracle_gram, deterministic=True)
    assert np.max(np.abs(Ad - A)) < 1e-9 and np.max(np.abs(Bd - B)) < 1e-9
    # fit()'s own product order on sharded data: P from the summed Gram, local (P G^T) Y, a second all-reduce
    Af, Bf = bd.fit_sharded(torch.from_numpy(Xb[b0:b1]), torch.from_numpy(Ub[b0:b1]), C, 1.0, 1e-3, gram_fn=_oracle_gram, order="fit",
                            apply_fn=_oracle_apply)
    assert np.max(np.abs(Af - A)) < 1e-7 and np.max(np.abs(Bf - B)) < 1e-7
    G1, Y1 = _oracle_gram(torch.from_numpy(Xb[b0:b1]), torch.from_numpy(Ub[b0:b1]), C, 1.0, b1 - b0, L, L + 1, L)
    G2, Y2 = G1.clone(), Y1.clone()
    bd.allreduce_gram_(G1, Y1, deterministic=True)
    bd.allreduce_gram_(G2, Y2, deterministic=True)
    assert torch.equal(G1, G2) and torch.equal(Y1, Y2)
    q.put((rank, A, B, Af, Bf))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fit_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), r[2]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, A0, B0, Af0, Bf0), (_, A1, B1, Af1, Bf1) = res
    assert np.array_equal(A0, A1) and np.array_equal(B0, B1)          # every rank solves the same reduced system
    assert np.array_equal(Af0, Af1) and np.array_equal(Bf0, Bf1)      # ... and holds the same all-reduced (P G^T) Y
    from oracle import edmdc_numpy as ek
    g = load_golden("edmdc.npz")
    X, U = g["X"], g["U"]
    Xl = [X[i * 200:(i + 1) * 200] for i in range(8)]
    Ul = [U[i * 200:(i + 1) * 200] for i in range(8)]
    A, B = ek.fit(Xl, Ul, g["centers"], 1.0, 1e-3)
    assert np.max(np.abs(A - A0)) < 1e-9 and np.max(np.abs(B - B0)) < 1e-9
    # the sharded fit-order result == the single-process (P G^T) Y on all bags
    GtG, _, _ = ek.gram(Xl, Ul, g["centers"], 1.0)
    P = np.linalg.pinv(GtG + 1e-3 * np.eye(GtG.shape[0]))
    M = _oracle_apply(torch.from_numpy(np.stack(Xl)), torch.from_numpy(np.stack([u[:-1] for u in Ul])), torch.from_numpy(g["centers"]), 1.0, 8, 199, 200, 199, P).numpy().T
    d = 12 + g["centers"].shape[0]
    assert np.max(np.abs(M[:, :d] - Af0)) < 1e-9 and np.max(np.abs(M[:, d:] - Bf0)) < 1e-9


# ---- sharded Lloyd (round 4): one integer all-reduce per iteration; the oracle's fixed-point stand-in runs the loop on the CPU ----
def _standin_lloyd(X_local, C0, mean, max_iter, tol_abs, allreduce, far_rows=None):
    from oracle import kmeans_numpy as kn
    ar = None if allreduce is None else (lambda buf, op: allreduce(torch.from_numpy(buf), op))      # shares the buffer's memory
    Xc = X_local.numpy() - (0.0 if mean is None else mean)
    shard = None
    if allreduce is not None:                              # this rank's place: rows of the ranks before it, rows of all
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(dist.get_world_size())]
        dist.all_gather(sizes, torch.tensor([len(Xc)], dtype=torch.int64))
        sizes = [int(t) for t in sizes]
        shard = (sum(sizes[:dist.get_rank()]), sum(sizes))
    C, labels, n_iter, nreloc = kn.lloyd_fixed_point(Xc, C0.numpy(), max_iter, tol_abs, allreduce=ar, shard=shard,
                                                     far_rows=far_rows or kn.far_rows_introselect)
    _standin_lloyd.relocations = nreloc
    return torch.from_numpy(C), torch.from_numpy(labels), 0.0, n_iter


def _km_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from bluerov2_dynamics_amd import dist as bd
        g = load_golden("edmdc.npz")
        X = g["X"][:1600]
        r0, r1 = bd.shard_range(len(X), rank, world)
        Xl = torch.from_numpy(X[r0:r1].copy())
        mean, var, Nt = bd.column_stats_sharded(Xl)
        assert Nt == len(X) and np.max(np.abs(mean.numpy() - X.mean(0))) < 1e-14 and abs(var - np.mean(np.var(X, axis=0))) < 1e-14
        C0 = torch.from_numpy(X[np.random.RandomState(1).choice(len(X), 24, replace=False)] - X.mean(0))
        C, labels, _, n_iter = bd.kmeans_lloyd_sharded(Xl, C0, mean=X.mean(0), max_iter=40, tol_abs=0.0, lloyd_fn=_standin_lloyd)
        q.put((rank, C.numpy(), labels.numpy(), n_iter))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))
        raise


def test_two_rank_lloyd_equals_single_process_bit_for_bit():
    """Sharded Lloyd under gloo, world size 2: every rank ends with the same centres, and they are the centres of the one-process
    run BIT FOR BIT (integer member sums: no order of additions to differ by); labels of the shards = labels of the whole."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_km_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), r[2]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import kmeans_numpy as kn
    g = load_golden("edmdc.npz")
    X = g["X"][:1600]
    C0 = X[np.random.RandomState(1).choice(len(X), 24, replace=False)] - X.mean(0)
    C1, lab1, it1, _ = kn.lloyd_fixed_point(X - X.mean(0), C0, 40, 0.0)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][1], C1) and res[0][3] == it1 == res[1][3]
    assert np.array_equal(np.concatenate([res[0][2], res[1][2]]), lab1)
    # ... and the fixed-point loop agrees with the floating-point restatement of scikit-learn's loop to rounding
    Cs, labs, _, its, _ = kn.lloyd(X - X.mean(0), C0, 40, 0.0)
    assert its == it1 and np.array_equal(labs, lab1) and np.max(np.abs(Cs - C1)) < 1e-13


# ---- an empty cluster in a sharded run (round 5): the far rows are chosen over ALL ranks' rows by the rule of the unsharded run ----------
def _reloc_case():
    g = load_golden("edmdc.npz")
    X = g["X"][:1600]
    C0 = X[np.random.RandomState(2).choice(len(X), 24, replace=False)].copy()
    C0[3] += 60.0; C0[11] -= 45.0; C0[20] += 25.0          # three initial centres far from all data: three empty clusters at once
    return X, C0


def _km_reloc_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from bluerov2_dynamics_amd import dist as bd
        from oracle import kmeans_numpy as kn
        X, C0 = _reloc_case()
        r0, r1 = bd.shard_range(len(X), rank, world)
        Xl = torch.from_numpy(X[r0:r1].copy())
        tol_abs = 1e-4 * float(np.mean(np.var(X, axis=0)))
        out = [rank]
        for rule in (kn.far_rows_numpy, kn.far_rows_introselect):
            fn = lambda *a, _r=rule: _standin_lloyd(*a, far_rows=_r)
            C, labels, _, n_iter = bd.kmeans_lloyd_sharded(Xl, torch.from_numpy(C0 - X.mean(0)), mean=X.mean(0), max_iter=300, tol_abs=tol_abs, lloyd_fn=fn)
            out += [C.numpy(), n_iter, _standin_lloyd.relocations]
        q.put(tuple(out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))
        raise


def test_two_rank_relocation_of_three_empty_clusters_equals_sklearn():
    """Three clusters run empty in the first iteration of a run sharded over two ranks.  With scikit-learn's own expression as the rule
    (np.argpartition on the gathered distances: what the Python layer's callback does) both ranks end with scikit-learn's centres and
    iteration count -- not merely with those of the one-rank run --; with the library's rule (NumPy's introselect restated) they end
    with the centres of the one-process run under that rule, bit for bit."""
    import warnings
    from sklearn.cluster import KMeans
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_km_reloc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), r[2]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import kmeans_numpy as kn
    X, C0 = _reloc_case()
    mean = X.mean(0)
    tol_abs = 1e-4 * float(np.mean(np.var(X, axis=0)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = KMeans(n_clusters=len(C0), init=C0, n_init=1, max_iter=300, tol=1e-4).fit(X)
    (_, Ca0, ita0, nra0, Cb0, itb0, nrb0), (_, Ca1, ita1, nra1, Cb1, itb1, nrb1) = res
    assert nra0 > 0 and nrb0 > 0
    assert np.array_equal(Ca0, Ca1) and np.array_equal(Cb0, Cb1)
    assert ita0 == ref.n_iter_ and np.max(np.abs(Ca0 + mean - ref.cluster_centers_)) < 1e-12           # scikit-learn's, on this host
    for rule, C2, it2 in ((kn.far_rows_numpy, Ca0, ita0), (kn.far_rows_introselect, Cb0, itb0)):
        C1, _, it1, nr1 = kn.lloyd_fixed_point(X - mean, C0 - mean, 300, tol_abs, far_rows=rule)
        assert nr1 > 0 and it1 == it2 and np.array_equal(C1, C2)                                           # the one-process run, same rule
