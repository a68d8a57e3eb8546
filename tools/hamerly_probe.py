"""How many samples would a bounds-based (Hamerly) Lloyd iteration have to re-scan on BASELINE config-3 data?  Simulation in
torch on the GPU box: exact top-2 distances every iteration (chunked fp64 matmul), the bounds evolve as the algorithm's would
(tightened only for samples that fail the test).  Prints per iteration: fraction failing the first test, fraction still failing
after tightening the upper bound (= the worklist of a full scan), label changes, centre shift.
    python3 tools/hamerly_probe.py [pairs] [iters]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, 512, 500
nb = max(1, pairs // L)
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
X = Xe.view(-1, n)
mean = X.mean(0)
X = X - mean
N = X.shape[0]
tol_abs = float(X.var(dim=0, unbiased=False).mean().item() * 1e-4)
C, _ = engine.kmeanspp_dev(Xe.view(-1, n), k, mean=mean.cpu().numpy(), random_state=0, ctx=ctx)
x2 = (X * X).sum(1)
R = float((2 * x2.max()).sqrt())
margin = 1e-6 * R


def top2(C):
    lab = torch.empty(N, dtype=torch.int64, device=dev); d1 = torch.empty(N, dtype=torch.float64, device=dev); d2 = torch.empty_like(d1)
    c2 = (C * C).sum(1)
    for i0 in range(0, N, 1 << 20):
        xs = X[i0:i0 + (1 << 20)]
        D = (x2[i0:i0 + (1 << 20), None] - 2 * xs @ C.T + c2[None]).clamp_min(0)
        v, ix = torch.topk(D, 2, dim=1, largest=False)
        lab[i0:i0 + (1 << 20)] = ix[:, 0]; d1[i0:i0 + (1 << 20)] = v[:, 0].sqrt(); d2[i0:i0 + (1 << 20)] = v[:, 1].sqrt()
    return lab, d1, d2


lab, ub, lb = top2(C)
print(f"N={N} k={k} tol_abs={tol_abs:.3e} R={R:.3f}", flush=True)
wave_any_total = 0.0
for it in range(1, iters + 1):
    sums = torch.zeros((k, n), dtype=torch.float64, device=dev).index_add_(0, lab, X)
    cnt = torch.zeros(k, dtype=torch.float64, device=dev).index_add_(0, lab, torch.ones(N, dtype=torch.float64, device=dev))
    Cn = torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1), C)
    delta = (Cn - C).norm(dim=1)
    shift = float((delta * delta).sum())
    C = Cn
    dsort, dix = delta.sort(descending=True)
    ub = ub + delta[lab]
    lb = lb - torch.where(lab == dix[0], dsort[1], dsort[0])
    fail1 = ~(ub + margin < lb)
    nl, d1, d2 = top2(C)                                   # ground truth
    dcur = (X - C[lab]).norm(dim=1)                        # exact distance to the current centre (the tightening step)
    ubt = torch.where(fail1, dcur, ub)
    fail2 = fail1 & ~(ubt + margin < lb)
    changed = int((nl != lab).sum())
    assert int(((nl != lab) & ~fail2).sum()) == 0, "a label changed outside the worklist: bounds are wrong"
    # waves (64 consecutive samples) with at least one failing lane -- what a non-compacted kernel would pay
    w1 = float(fail1.view(-1)[: N // 64 * 64].view(-1, 64).any(1).double().mean())
    w2 = float(fail2.view(-1)[: N // 64 * 64].view(-1, 64).any(1).double().mean())
    ub = torch.where(fail2, d1, ubt); lb = torch.where(fail2, d2, lb); lab = torch.where(fail2, nl, lab)
    if it <= 10 or it % 10 == 0:
        print(f"it {it:3d}: fail1 {float(fail1.double().mean()):.4f} (waves {w1:.3f})  worklist {float(fail2.double().mean()):.4f} (waves {w2:.3f})  changed {changed}  shift {shift:.3e}", flush=True)
    wave_any_total += float(fail2.double().mean())
    if shift <= tol_abs or changed == 0:
        print("converged at", it); break
print(f"mean worklist fraction over {it} iterations: {wave_any_total / it:.4f}")
