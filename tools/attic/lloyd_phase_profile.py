"""Where does a wave of the filtered E-step spend its time?  Needs a library built with -DKM_PROFILE=1 (kmeans.hip: s_memtime stamps
at the phase boundaries, summed per wave into a device array) selected through BROV2_LIBRARY:
    python tools/build_variants.py kmeans.hip:prof=-DKM_PROFILE=1
    BROV2_LIBRARY=$PWD/build_variants/prof/libbrov2.so python3 tools/lloyd_phase_profile.py [pairs] [iters]
Prints the share of wave time per phase (rows load | label groups | candidate masks | candidate evaluation | labels + member sums)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
lib = _lib.load_library()
prof = lib.brov_debug_kmprof
prof.restype = ctypes.c_int
prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
names = ["rows load", "label groups", "candidate masks", "evaluation", "labels + member sums"]
exp = bool(lib.brov_experiments_build())
runs = ((0, "default: sorted order, packed fp32 screening, distance bounds"), (4, "distance bounds off"))
if exp:
    runs += ((128, "sorted, screening off (fp64 evaluation of every candidate)"),)
if len(sys.argv) > 3:
    runs += ((2, "caller's order"),)
for variant, label in runs:
    ctx.set_kmeans_variant(variant)
    tm = {}
    ctx.set_timing(True)
    prof(buf, 1)
    C, inertia, n_iter = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=iters, ctx=ctx, timings=tm)
    torch.cuda.synchronize()
    prof(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    tot = v[:5].sum()
    print(f"{label}: {n_iter} iterations, Lloyd {tm['lloyd_ms']:.1f} ms; {int(v[7])} wave passes, {tot / v[7]:.0f} ticks per pass", flush=True)
    if variant == 64:
        print("    ticks per pass: rows+x %.0f | reference, radius, prefix %.0f | packed loop %.0f | exact pair %.0f | stores + member sums %.0f" % tuple(v[q] / max(v[7], 1) for q in range(5)))
        print(f"    pk kernel: {int(v[7])} passes, {int(v[8])} screened ({v[9] / max(v[8], 1):.1f} candidates), {int(v[10])} not certified, {int(v[13])} full scans")
        continue
    if v[14]:
        print(f"    settled by the packed-fp32 screening: {int(v[14])} passes ({v[15] / v[14]:.1f} candidates)")
    print(f"    single-reference passes {int(v[8])} ({v[9] / max(v[8], 1):.1f} candidates, {int(v[10])} repeated for a tie), mask-form passes {int(v[11])} "
          f"({v[12] / max(v[11], 1):.1f} candidates), full scans {int(v[13])}")
    for nm, t in zip(names, v[:5]):
        print(f"    {nm:22s} {100 * t / tot:5.1f} %   {t / v[7]:8.0f} ticks per wave pass")
