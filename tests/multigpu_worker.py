"""One rank of the multi-GPU rehearsal (child process of tests/test_multigpu.py; one process per GPU).

    python tests/multigpu_worker.py MODE RANK WORLD RENDEZVOUS OUT [TOTAL_ROLLOUTS] [HORIZON] [BOUNDS_RATE] [KMEANS_K]

MODE "torch": torch.distributed (backend nccl = RCCL), RENDEZVOUS = TCP port on 127.0.0.1.
MODE "gloo" : torch.distributed over gloo, every rank on the SAME GPU (cuda:RANK mod the device count) -- the rehearsal a one-GPU box allows.
MODE "brov" : the torch-free communicator of the C ABI (brov_comm_unique_id / brov_comm_init_rank /
              edmdc_gram_allreduce_dev); RENDEZVOUS = a file path: rank 0 writes the 128-byte id there, the others wait for it.
BASELINE config 4 at small size: TOTAL_ROLLOUTS x HORIZON RK4 steps in total, contiguous shards (dist.shard_range), commands
a function of the GLOBAL trajectory index, trajectories stored [B][T+1][12], centres = seeded rows of the first 64 trajectories
(which every rank rolls out for itself: no exchange), local lift + G^T[G|Y], all-reduce, identical host solve; both product
orders (fit_multi: one collective; fit: a second one for the p x d block).  Writes OUT (npz): GtG, GtY, A, B, Af, Bf.
The GPU is chosen (cuda:RANK) before anything initialises HIP in this process."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    mode, rank, world, rdv, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    Bt = int(sys.argv[6]) if len(sys.argv) > 6 else 1536
    T = int(sys.argv[7]) if len(sys.argv) > 7 else 40
    bounds_rate = float(sys.argv[8]) if len(sys.argv) > 8 else None       # edmdc_set_kmeans_bounds_rate (1.0: list form from the first sorted iteration)
    km_k = int(sys.argv[9]) if len(sys.argv) > 9 else 48                  # clusters of the k-means leg (>= 64: candidate filter, sorted order, bounds)
    import torch
    ndev = torch.cuda.device_count()                    # does not initialise the GPU on this image
    dev_id = rank % max(ndev, 1)
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    from bluerov2_dynamics_amd import _lib, engine
    from bluerov2_dynamics_amd import dist as bd
    ctx = _lib.default_context(dev_id)
    if bounds_rate is not None:
        ctx.set_kmeans_bounds_rate(bounds_rate)
    n, r, k, gamma, ridge, dt = 12, 8, 48, 1.0, 1e-3, 0.02

    def rollouts(b0, nb):
        U = torch.empty((nb, T, r), dtype=torch.float64, device=dev)
        engine.fill_controls_dev(U, "btu", "ar1", seed=0xC0F4, b0=b0, T_total=T, ctx=ctx)
        X = torch.empty((nb, T + 1, n), dtype=torch.float64, device=dev)
        x0 = torch.zeros((nb, n), dtype=torch.float64, device=dev)
        x0[:, 2] = 5.0
        engine.rollout_dev(_lib.THRUSTER_EULER, "rk4", x0, U, dt, traj=X, layout="btu", ctx=ctx)
        return X, U

    Xc, _ = rollouts(0, 64)
    idx = torch.from_numpy(np.random.RandomState(0).choice(64 * (T + 1), k, replace=False)).to(dev)
    idxk = idx if km_k == k else torch.from_numpy(np.random.RandomState(0).choice(64 * (T + 1), km_k, replace=False)).to(dev)
    Cdup = Xc.view(-1, n)[idxk].contiguous()            # any rows: several are the common initial state -- duplicate centres (k-means leg)
    idx1 = torch.from_numpy(np.random.RandomState(0).choice(64 * T, k, replace=False)).to(dev)
    C = Xc[:, 1:, :].reshape(-1, n)[idx1].contiguous()  # rows after the first step: distinct centres, a well-conditioned Gram (fit legs)
    b0, b1 = bd.shard_range(Bt, rank, world)
    X, U = rollouts(b0, b1 - b0)

    comm = None
    if mode in ("torch", "gloo"):
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=rdv, RANK=str(rank), WORLD_SIZE=str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if mode == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            allreduce = None
        else:
            # several ranks on ONE GPU (RCCL refuses that; the one-GPU development box): gloo, the blocks staged through the host
            dist.init_process_group("gloo", rank=rank, world_size=world)

            def allreduce(a_, b_):
                for t_ in (a_, b_):
                    h_ = t_.cpu()
                    dist.all_reduce(h_)
                    t_.copy_(h_)
                return a_, b_
    else:
        if rank == 0:
            ident = _lib.Comm.unique_id()
            with open(rdv + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(rdv + ".tmp", rdv)
        else:
            t0 = time.time()
            while not os.path.exists(rdv):
                if time.time() - t0 > 120:
                    raise TimeoutError("no communicator id from rank 0")
                time.sleep(0.02)
            ident = open(rdv, "rb").read()
        comm = _lib.Comm(dev_id, ident, world, rank)
        allreduce = comm.allreduce_gram_

    # the exchange on its own (the summed blocks are part of the output), then both product orders end to end
    GtG, GtY = bd._device_gram(X, U, C, gamma, b1 - b0, T, T + 1, T)
    if allreduce is None:
        bd.allreduce_gram_(GtG, GtY)
    else:
        allreduce(GtG, GtY)
    A, B = bd.fit_sharded(X, U, C, gamma, ridge, allreduce=allreduce)
    Af, Bf = bd.fit_sharded(X, U, C, gamma, ridge, order="fit", allreduce=allreduce)
    # round 4: the centres over ALL ranks' states (Koopman/koopmanEDMDc.py:85 runs KMeans over all of X): Lloyd's loop with one integer
    # all-reduce per iteration, from common initial centres (the seeded rows above, centred with the column means of the first 64
    # trajectories, which every rank holds) -- the same bits on every rank and for every world size
    mean_h = Xc.view(-1, n).mean(dim=0).cpu().numpy()
    C0 = Cdup - torch.from_numpy(mean_h).to(dev)       # duplicate centres: empty clusters in the first iteration -> the sharded relocation
    if comm is not None:
        ctx.kmeans_use_comm(comm)
        ctx.set_kmeans_shard(rank, world, b0 * (T + 1), Bt * (T + 1))
    # (round 5) a sharded run applies the ctx's selection rule to the distances of ALL ranks' rows: first the library's own rule
    # (NumPy's introselect restated: what a plain-C caller gets), then the default -- np.argpartition on this host through the callback
    ctx.set_kmeans_far_select(False)
    Ck_lib, _, _, iters_lib = bd.kmeans_lloyd_sharded(X.view(-1, n), C0, mean=mean_h, max_iter=25, tol_abs=0.0, ctx=ctx)
    reloc_lib = ctx.kmeans_relocations()
    ctx.set_kmeans_far_select(True)
    Ck, labels_k, inertia_k, iters_k = bd.kmeans_lloyd_sharded(X.view(-1, n), C0, mean=mean_h, max_iter=25, tol_abs=0.0, ctx=ctx)
    loop_info = ctx.kmeans_loop_info()
    forced = False
    if mode == "torch" and world == 1:
        # one rank, backend nccl: the same loop once more with the exchange installed -- every all-reduce of the loop goes through RCCL
        # on the library's own device words (the path the N-rank runs take), and must change nothing
        Cf, labels_f, inertia_f, iters_f = bd.kmeans_lloyd_sharded(X.view(-1, n), C0, mean=mean_h, max_iter=25, tol_abs=0.0, ctx=ctx, force_exchange=True)
        assert torch.equal(Cf, Ck) and torch.equal(labels_f, labels_k) and iters_f == iters_k and inertia_f == inertia_k
        forced = True
    # ... and the k-means++ seeding over all ranks' rows: scikit-learn's random stream for Bt (T + 1) rows, two small exchanges per centre
    if comm is not None:
        Cs, idx_s = engine.kmeanspp_dev(X.view(-1, n), km_k, mean=mean_h, random_state=3, ctx=ctx, n_global=Bt * (T + 1))
    else:
        Cs, idx_s = bd.kmeanspp_sharded(X.view(-1, n), km_k, mean=mean_h, random_state=3, ctx=ctx)
    if comm is not None:
        ctx.kmeans_use_comm(None)
        ctx.set_kmeans_shard()
    torch.cuda.synchronize()
    np.savez(out, GtG=GtG.cpu().numpy(), GtY=GtY.cpu().numpy(), A=A, B=B, Af=Af, Bf=Bf, b0=b0, b1=b1, device=dev_id,
             Ck=Ck.cpu().numpy(), iters_k=iters_k, inertia_k=inertia_k, labels_k=labels_k.cpu().numpy(), reloc_k=ctx.kmeans_relocations(),
             Cs=Cs.cpu().numpy(), idx_s=idx_s, forced_exchange=forced, Ck_lib=Ck_lib.cpu().numpy(), iters_lib=iters_lib, reloc_lib=reloc_lib,
             ridge=ridge, resorts=loop_info["resorts"], first_resort=loop_info["first_resort_iteration"], list_e_steps=loop_info["list_form_e_steps"])
    if mode in ("torch", "gloo"):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    else:
        comm.close()


if __name__ == "__main__":
    main()
