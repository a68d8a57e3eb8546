"""How much of the Lloyd E-step could distance bounds skip?  (probe; torch only, statistics not bits)

Simulates Hamerly's two bounds per sample (upper: distance to its centre, lower: distance to the second closest centre, both moved
by the centres' shifts after every M-step) on BASELINE config-3 data (1e7 states, k = 512) over the iterations of the loop, with the
lower bound capped by what a scan of the centres within 2 (1 + beta) r of the sample's own centre can certify ((1 + 2 beta) r):
    python3 tools/hamerly_probe.py [iters] [N]
Prints per iteration: labels changed, samples whose bounds fail before / after tightening the upper bound, per beta."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
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
mean = X.mean(dim=0)
C, idx = engine.kmeanspp_dev(X, k, mean=mean.cpu().numpy(), random_state=0, ctx=ctx)
Xc = (X - mean).contiguous()
C = C.clone()
N = Xc.shape[0]
x2 = (Xc * Xc).sum(dim=1)


SECOND = torch.empty(Xc.shape[0], dtype=torch.int64, device=dev)      # id of the second closest centre, distance to the third (last scan)
THIRD = torch.empty(Xc.shape[0], dtype=torch.float64, device=dev)


def scan(C):
    """labels, distance to the closest and second closest centre"""
    lab = torch.empty(N, dtype=torch.int64, device=dev)
    d1 = torch.empty(N, dtype=torch.float64, device=dev)
    d2 = torch.empty(N, dtype=torch.float64, device=dev)
    c2 = (C * C).sum(dim=1)
    for s in range(0, N, 1 << 20):
        D = (x2[s:s + (1 << 20), None] + c2[None, :] - 2.0 * (Xc[s:s + (1 << 20)] @ C.T)).clamp_min_(0.0)
        v, i = torch.topk(D, 3, dim=1, largest=False)
        lab[s:s + (1 << 20)] = i[:, 0]
        d1[s:s + (1 << 20)] = v[:, 0].sqrt()
        d2[s:s + (1 << 20)] = v[:, 1].sqrt()
        SECOND[s:s + (1 << 20)] = i[:, 1]
        THIRD[s:s + (1 << 20)] = v[:, 2].sqrt()
    return lab, d1, d2


betas = (0.0, 0.1, 0.25, 0.5, 1e9)
TOP = int(os.environ.get("PROBE_TOP", "0"))
KNS = [int(v) for v in os.environ.get("PROBE_KNS", "").split(",") if v]
KN = int(os.environ.get("PROBE_KN", "0"))       # > 0: the lower bound decays by the largest shift among the KN nearest centres of the sample's own
                                                # (the others are held off by the triangle inequality through the own centre) instead of the largest of all
lab, d1, d2 = scan(C)
ub = {b: d1.clone() for b in betas}
lb = {b: torch.minimum(d2, (1.0 + 2.0 * b) * d1) for b in betas}
# two lower bounds (PROBE_ELKAN=1): l1 for the runner-up alone (gives way by ITS shift), l2 for all the others (by the largest shift)
ELKAN = os.environ.get("PROBE_ELKAN") == "1"
if ELKAN:
    e_u, e_l1, e_b, e_l2 = d1.clone(), d2.clone(), SECOND.clone(), torch.minimum(THIRD, 1.12 * d1 + 0 * d1 + 1e9)
print("iter changed%  " + "  ".join(f"b={b:g}: fail% / after-tighten%" for b in betas), flush=True)
for it in range(1, iters + 1):
    sums = torch.zeros((k, n), dtype=torch.float64, device=dev).index_add_(0, lab, Xc)
    cnt = torch.bincount(lab, minlength=k).clamp_min(1).to(torch.float64)
    Cn = sums / cnt[:, None]
    p = (Cn - C).norm(dim=1)
    C = Cn
    top = torch.topk(p, 2).values
    pa = p[lab]
    other = torch.where(pa == top[0], top[1], top[0])            # largest shift among the other centres
    far = None
    multi = []
    if KN > 0:
        Dn, In = torch.sort(torch.cdist(C, C), dim=1)
        m_near = p[In[:, 1:KN + 1]].max(dim=1).values            # per own centre: the largest shift among its KN nearest
        other = m_near[lab]
        far = Dn[:, KN + 1][lab]                                 # distance from the own centre to the first centre outside that set
    if KNS:
        # several near sets at once (each gives a valid bound; the sample takes the best), the set of ALL centres among them
        Dn, In = torch.sort(torch.cdist(C, C), dim=1)
        for kn in KNS:
            multi.append((p[In[:, 1:kn + 1]].max(dim=1).values[lab], Dn[:, kn + 1][lab]))
    top_rule = None
    if TOP > 0:
        # the TOP largest shifts apart: such a mover counts for a sample only if it is near the sample's own centre
        tv, ti = torch.topk(p, TOP + 1)
        pr = p.clone(); pr[ti[:TOP]] = 0.0
        m_rest = pr.max()                                           # largest shift among the other centres
        Dm = torch.cdist(C, C[ti[:TOP]])                            # [k][TOP] distances from every centre to the movers
        top_rule = (m_rest, tv[:TOP], ti[:TOP], Dm)
    lab_n, d1, d2 = scan(C)
    changed = (lab_n != lab).float().mean().item() * 100
    da = (Xc - C[lab]).norm(dim=1)                               # exact distance to the old centre (tightening)
    line = f"{it:4d} {changed:7.3f}   "
    for b in betas:
        u = ub[b] + pa
        l = lb[b] - other
        if far is not None:
            l = torch.minimum(l, far - u)
        for (mk, fk) in multi:
            l = torch.maximum(l, torch.minimum(lb[b] - mk, fk - u))
        if top_rule is not None:
            m_rest, tv, ti, Dm = top_rule
            l = lb[b] - m_rest
            for t in range(TOP):
                cand = torch.maximum(lb[b] - tv[t], Dm[lab, t] - u)
                cand = torch.where(lab == ti[t], torch.full_like(cand, float("inf")), cand)
                l = torch.minimum(l, cand)
        f1 = u >= l
        f2 = f1 & (da >= l)
        line += f"  {f1.float().mean().item() * 100:6.2f} / {f2.float().mean().item() * 100:6.2f}"
        # a skipped sample must keep its label
        wrong = ((~f2) & (lab_n != lab)).sum().item()
        if wrong:
            line += f" (!{wrong})"
        ub[b] = torch.where(f2, d1, torch.where(f1, da, u))
        lb[b] = torch.where(f2, torch.minimum(d2, (1.0 + 2.0 * b) * d1), l)
    if ELKAN:
        u = e_u + pa
        l1 = e_l1 - p[e_b]
        l2 = e_l2 - top[0]
        f1 = u >= torch.minimum(l1, l2)
        wrong = ((~f1) & (lab_n != lab)).sum().item()
        line += f"   two bounds: fail {f1.float().mean().item() * 100:6.2f} %" + (f" (!{wrong})" if wrong else "")
        e_u = torch.where(f1, d1, u)
        e_l1 = torch.where(f1, d2, l1)
        e_b = torch.where(f1, SECOND, e_b)
        e_l2 = torch.where(f1, THIRD, l2)
    print(line + f"   max shift {top[0].item():.2e}  median shift {p.median().item():.2e}  median r {d1.median().item():.2e}", flush=True)
    lab = lab_n
# candidates a scan radius costs: centres within 2 (1 + beta) r of the sample's own centre
Dcc = torch.cdist(C, C)
sel = torch.randint(0, N, (200000,), device=dev)
rr = d1[sel]
row = Dcc[lab[sel]]
print("candidates within 2 (1 + beta) r:", {b: round((row < (2.0 * (1.0 + b) * rr)[:, None]).sum(dim=1).float().mean().item(), 1) for b in betas[:-1]})
