"""Child process of tests/test_gpu_parity.py::test_dropin_classes_* : every method of the drop-in classes a script of the reference
calls (KoopmanEDMDc.fit / fit_multi / evaluate / multistep_rmse / simulate / _lift, the three BlueROV2 classes' dynamics(),
compute_thruster_forces(), simulate-style loops through engine.rollout) on the committed fixtures, results to an .npz -- so that
the parent can compare a process that never imports torch with one that runs the torch-tensor path, bit for bit.

    python tests/dropin_worker.py <out.npz> <arrays: native|torch>

TEST INFRASTRUCTURE; the oracle is not used here (the parent compares the two product runs and the reference fixtures)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def main():
    out_path, arrays = sys.argv[1], sys.argv[2]
    from bluerov2_dynamics_amd import _lib, engine
    from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
    from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2
    from bluerov2_dynamics_amd.fossen.BlueROV2_thrust import BlueROV2 as BlueROV2Thrust
    from bluerov2_dynamics_amd.fossen.BlueROV2_wrench import BlueROV2 as BlueROV2Wrench
    out = {}
    g = np.load(os.path.join(GOLDEN, "edmdc.npz"))
    X, U, nt = g["X"], g["U"], int(g["n_train"])
    k, gamma, ridge = int(g["k"]), float(g["gamma"]), float(g["ridge"])
    Xt, Ut = X[nt:], U[nt:]
    # fit with the class's own centres (device k-means++ / Lloyd), then the scores
    m = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge, arrays=arrays)
    m.fit(X[:nt], U[:nt])
    out.update(fit_centers=m.centers_, fit_A=m.A_, fit_B=m.B_, fit_eval=m.evaluate(Xt, Ut),
               fit_ms=np.array([m.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)]), fit_sim=m.simulate(Xt[0], Ut[:50]),
               fit_lift=m._lift(Xt[:7]))
    # fit with the reference's centres: comparable with the reference's A, B
    m.fit(X[:nt], U[:nt], centers=g["centers"])
    out.update(refc_A=m.A_, refc_B=m.B_, refc_ms=np.array([m.multistep_rmse(Xt, Ut, H) for H in (1, 10, 100)]))
    cuts = [(0, 500), (500, 501), (501, 1300), (1300, 1600)]
    m2 = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=k, gamma=gamma, ridge=ridge, arrays=arrays)
    m2.fit_multi([X[a:b] for a, b in cuts], [U[a:b] for a, b in cuts])
    out.update(multi_centers=m2.centers_, multi_A=m2.A_, multi_B=m2.B_)
    # the vehicles: three stateful calls each, as the scripts' loops make them
    kat = np.load(os.path.join(GOLDEN, "fossen_rhs_kat.npz"))
    rov = BlueROV2(current_speed=kat["thr_cur_cur"].copy())
    out["thr_xdot"] = np.stack([rov.dynamics(kat["thr_cur_X"][5], kat["thr_cur_U"][5], float(kat["thr_cur_dt"])) for _ in range(3)])
    out["thr_lag"] = np.stack([l._x for l in rov.thruster_lags])
    out["thr_tau"] = BlueROV2().compute_thruster_forces(kat["thr_U"][5], float(kat["thr_dt"]))
    out["we_xdot"] = BlueROV2Thrust(current_speed=kat["we_cur_cur"].copy()).dynamics(kat["we_cur_X"][5], kat["we_cur_U"][5])
    out["wq_xdot"] = BlueROV2Wrench(current_speed=kat["wq_cur_cur"].copy()).dynamics(kat["wq_cur_X"][5], kat["wq_cur_U"][5])
    # a batched rollout and the windowed evaluator through the host entry points
    rng = np.random.default_rng(3)
    x0 = np.zeros((64, 12))
    x0[:, 2] = 5.0
    Ur = rng.uniform(-1, 1, (64, 120, 8))
    out["rollout_xT"] = engine.rollout(_lib.THRUSTER_EULER, "rk4", x0, Ur, 0.02)["xT"]
    w = np.load(os.path.join(GOLDEN, "windows.npz"))
    out["window_rmse"] = np.array([engine.window_rmse(_lib.THRUSTER_EULER, "euler", w["X"], w["U"], int(H), float(w["dt"])) for H in (1, 10)])
    out["torch_imported"] = np.array("torch" in sys.modules)
    out["hip_runtime"] = np.array(_lib.hip_runtime)
    np.savez(out_path, **out)


if __name__ == "__main__":
    main()
