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
def _standin_lloyd(X_local, C0, mean, max_iter, tol_abs, allreduce):
    from oracle import kmeans_numpy as kn
    ar = None if allreduce is None else (lambda buf, op: allreduce(torch.from_numpy(buf), op))      # shares the buffer's memory
    Xc = X_local.numpy() - (0.0 if mean is None else mean)
    C, labels, n_iter, _ = kn.lloyd_fixed_point(Xc, C0.numpy(), max_iter, tol_abs, allreduce=ar)
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
