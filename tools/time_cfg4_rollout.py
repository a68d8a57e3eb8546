"""BASELINE config 4's rollout leg at one GPU: 2^20 trajectories x 500 RK4 steps in the caller layout [B][T][8] / [B][T+1][12]:
the AR(1) command fill and the rollout, two-wave kernel against the one-lane kernel (brov_set_rollout_variant 1) and the
LDS-staged one.  Run on the GPU box:  python3 tools/time_cfg4_rollout.py [B] [T]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
U = torch.empty((B, T, 8), dtype=torch.float64, device=dev)
X = torch.empty((B, T + 1, 12), dtype=torch.float64, device=dev)
x0 = torch.zeros((B, 12), dtype=torch.float64, device=dev); x0[:, 2] = 5.0


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    return min(e[i].elapsed_time(e[i + 1]) for i in range(reps))


ctx = _lib.default_context(0)
for dist in ("ar1", "iid"):
    ms = timed(lambda: engine.fill_controls_dev(U, "btu", dist, seed=0xC0F4, b0=0, T_total=T, ctx=ctx))
    print(f"fill {dist} BTU: {ms:.2f} ms = {B * T * 64 / ms / 1e9:.2f} TB/s written", flush=True)
engine.fill_controls_dev(U, "btu", "ar1", seed=0xC0F4, b0=0, T_total=T, ctx=ctx)
ref = None
for name, env, mode in (("two-wave, lane-per-row", None, 0), ("one-lane, lane-per-row", "1", 2), ("one-lane, LDS-staged", "1", 1)):
    c = _lib.Context(0)
    c.set_rollout_variant(1 if env else 0)
    c.set_btu_staging(mode)
    for integ in ("rk4", "euler"):
        for store in (True, False):
            ms = timed(lambda: engine.rollout_dev(_lib.THRUSTER_EULER, integ, x0, U, 0.02, traj=X if store else None, layout="btu", ctx=c))
            print(f"{name:24s} {integ:5s} {'stored' if store else 'endpoint':8s}: {ms:8.2f} ms = {B * T / ms / 1e6:.3f}e9 steps/s, "
                  f"{B * T * (160 if store else 64) / ms / 1e9:.2f} TB/s", flush=True)
    c.close()
