"""Pin the CPU oracle (oracle/) against the fixtures generated from the imported reference
(tools/gen_golden.py).  CPU only.  Tolerances: 1e-12 relative (the reference's BLAS-internal
summation order is not reproducible bit for bit; see oracle/brov2_oracle.c header)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import controls, edmdc_numpy as ek, fossen_c as fc

TOL = 1e-12


def test_constants_and_discretisation():
    g = load_golden("fossen_constants.npz")
    c = fc.constants()
    assert rel_err(c["Minv"], np.diag(g["Minv"])) < 1e-15
    assert rel_err(c["alloc"], g["alloc"]) < 1e-15
    assert rel_err(c["thr_r"], g["thr_r"]) < 1e-15
    assert rel_err(c["thr_dir"], g["thr_dir"]) < 1e-15
    for dt in g["dts"]:
        Ad, Bd = fc.discretise_lag(float(dt))
        assert rel_err(Ad, g[f"Ad_{dt}"]) < 1e-14
        assert rel_err(Bd, g[f"Bd_{dt}"]) < 1e-14


@pytest.mark.parametrize("tag", ["thr", "thr_cur"])
def test_thruster_rhs_three_stateful_calls(tag):
    g = load_golden("fossen_rhs_kat.npz")
    X, U, dt, cur = g[f"{tag}_X"], g[f"{tag}_U"], float(g[f"{tag}_dt"]), g[f"{tag}_cur"]
    lag = None
    for c in range(3):
        xd, lag = fc.rhs(fc.MODEL_THRUSTER_EULER, X, U, dt, lag=lag, current=cur)
        ref = g[f"{tag}_XDOT"][c]
        fin = np.isfinite(ref)
        # rows 0,1 are the theta = +-pi/2 clamp cases: tan ~ 1e7 amplifies the last ulp of cos
        assert np.array_equal(np.isfinite(xd), fin)
        assert rel_err(xd[2:], ref[2:]) < TOL
        assert np.max(np.abs(xd[:2] - ref[:2]) / np.maximum(1.0, np.abs(ref[:2]))) < 1e-9
        assert rel_err(lag, g[f"{tag}_LAG"][c]) < TOL
    tau, _ = fc.thruster_forces(U, dt)
    assert rel_err(tau, g[f"{tag}_TAU1"]) < TOL


@pytest.mark.parametrize("tag,model", [("we", fc.MODEL_WRENCH_EULER), ("we_cur", fc.MODEL_WRENCH_EULER),
                                       ("wq", fc.MODEL_WRENCH_QUAT), ("wq_cur", fc.MODEL_WRENCH_QUAT)])
def test_wrench_rhs(tag, model):
    g = load_golden("fossen_rhs_kat.npz")
    xd, _ = fc.rhs(model, g[f"{tag}_X"], g[f"{tag}_U"], 0.02, current=g[f"{tag}_cur"])
    ref = g[f"{tag}_XDOT"]
    lo = 2 if model == fc.MODEL_WRENCH_EULER else 0
    assert rel_err(xd[lo:], ref[lo:]) < TOL
    assert np.max(np.abs(xd[:lo] - ref[:lo]) / np.maximum(1.0, np.abs(ref[:lo])), initial=0.0) < 1e-9


def test_survey_kats():
    """Compact KATs quoted in SURVEY.md section 8(a)."""
    x = np.array([0.3, -0.2, 1.0, 0.1, -0.2, 0.7, 0.4, -0.3, 0.2, 0.05, -0.1, 0.2])
    tau = np.array([10, -5, 3, 0.5, -0.4, 0.8])
    xd, _ = fc.rhs(fc.MODEL_WRENCH_EULER, x, tau)
    ref = [0.47931379506139227, -0.01266340763915013, 0.24514877925621098, 0.01168425760838594,
           -0.11946709985716822, 0.19286188883949557, -0.9284243059999626, 0.6478701364644579,
           -0.33335869293622805, 2.375088715763591, 2.556589038374261, 1.403344594594595]
    assert rel_err(xd[0], ref) < 1e-14
    u = np.array([0.1, -0.2, 0.3, -0.4, 0.5, -0.6, 0.7, -0.8])
    xd1, lag = fc.rhs(fc.MODEL_THRUSTER_EULER, x, u, 0.02)
    xd2, _ = fc.rhs(fc.MODEL_THRUSTER_EULER, x, u, 0.02, lag=lag)
    assert rel_err(xd1[0, 6:], [-1.3696415288010468, 0.41406407349169017, -0.27971576990592095,
                                -24.803789487103767, 3.42884851717307, 2.8145185134400537]) < 1e-13
    assert rel_err(xd2[0, 6:], [-1.3336320564242452, 0.13880134287327764, -0.19483573957649727,
                                -39.867757441222288, 3.2996050336711238, 4.4110704113983985]) < 1e-13


def test_cfg2_rollouts_rk4_and_euler():
    g = load_golden("fossen_rollouts.npz")
    T, dt, sub, seed = int(g["cfg2_T"]), float(g["cfg2_dt"]), int(g["cfg2_sub"]), int(g["cfg2_seed"])
    U = controls.controls_iid(seed, 0, 8, T)
    assert np.array_equal(U[:, :4, :], g["cfg2_U_head"])
    x0 = np.tile(g["cfg2_x0"], (8, 1))
    r = fc.rollout(fc.MODEL_THRUSTER_EULER, fc.INTEG_RK4, x0, U, dt, sub=sub, nthreads=8)
    assert rel_err(r["traj"], g["cfg2_rk4"]) < 1e-10
    assert rel_err(r["lag"], g["cfg2_rk4_lag_end"]) < 1e-10
    e = fc.rollout(fc.MODEL_THRUSTER_EULER, fc.INTEG_EULER, x0, U, dt, sub=sub, nthreads=8)
    assert rel_err(e["traj"], g["cfg2_euler"]) < 1e-10
    assert rel_err(e["lag"], g["cfg2_euler_lag_end"]) < 1e-10
    assert rel_err(e["xT"], g["cfg2_euler"][:, -1]) < 1e-10


def test_ar1_wrench_quat_and_cfg1_rollouts():
    g = load_golden("fossen_rollouts.npz")
    dt, sub = float(g["ar1_dt"]), int(g["ar1_sub"])
    for integ, key in ((fc.INTEG_RK4, "ar1_rk4"), (fc.INTEG_EULER, "ar1_euler")):
        r = fc.rollout(fc.MODEL_THRUSTER_EULER, integ, g["ar1_X0"], g["ar1_U"], dt, sub=sub)
        assert rel_err(r["traj"], g[key]) < 1e-10
    dt, sub = float(g["w_dt"]), int(g["w_sub"])
    for model, integ, x0k, key in ((fc.MODEL_WRENCH_EULER, fc.INTEG_EULER, "we_X0", "we_euler"),
                                   (fc.MODEL_WRENCH_EULER, fc.INTEG_RK4, "we_X0", "we_rk4"),
                                   (fc.MODEL_WRENCH_QUAT, fc.INTEG_EULER, "wq_X0", "wq_euler"),
                                   (fc.MODEL_WRENCH_QUAT, fc.INTEG_RK4, "wq_X0", "wq_rk4_ext")):
        r = fc.rollout(model, integ, g[x0k], g["w_TAU"], dt, sub=sub)
        assert rel_err(r["traj"], g[key]) < 1e-10, key
    x0 = np.zeros((1, 12))
    x0[0, 2] = 5.0
    U = np.tile(g["cfg1_u"], (1, 1000, 1))
    r = fc.rollout(fc.MODEL_THRUSTER_EULER, fc.INTEG_EULER, x0, U, 0.02, sub=10)
    assert rel_err(r["traj"][0], g["cfg1_euler"]) < 1e-11


def test_window_rmse_carried_lag():
    g = load_golden("windows.npz")
    X, U, TAU, Xq, dt = g["X"], g["U"], g["TAU"], g["Xq"], float(g["dt"])
    for i, H in enumerate(g["H"]):
        H = int(H)
        assert abs(fc.window_rmse(fc.MODEL_THRUSTER_EULER, fc.INTEG_EULER, X, U, H, dt) - g["thr_euler_rmse"][i]) < 1e-12
        assert abs(fc.window_rmse(fc.MODEL_THRUSTER_EULER, fc.INTEG_RK4, X, U, H, dt) - g["thr_rk4_rmse"][i]) < 1e-12
        assert abs(fc.window_rmse(fc.MODEL_WRENCH_EULER, fc.INTEG_EULER, X, TAU, H, dt) - g["we_euler_rmse"][i]) < 1e-12
        assert abs(fc.window_rmse(fc.MODEL_WRENCH_QUAT, fc.INTEG_EULER, Xq, TAU, H, dt) - g["wq_euler_rmse"][i]) < 1e-12
    # quirk Q2 is real: a fresh lag per window gives a different number
    fresh = fc.window_rmse(fc.MODEL_THRUSTER_EULER, fc.INTEG_EULER, X, U, 10, dt, carry_lag=False)
    assert abs(fresh - g["thr_euler_rmse"][1]) > 1e-6


def test_double_integrator_baseline():
    g, w = load_golden("di.npz"), load_golden("windows.npz")
    X, U, TAU, Xq, dt = w["X"], w["U"], w["TAU"], w["Xq"], float(w["dt"])
    Kl, Ka = fc.estimate_di_gains(X[:300], U[:300], dt)
    assert rel_err(Kl, g["thr_Klin"]) < 1e-14 and rel_err(Ka, g["thr_Kang"]) < 1e-14
    fc.set_di_gains(g["thr_Klin"], g["thr_Kang"])
    assert rel_err(fc.rollout(fc.MODEL_DI_THRUSTER_EULER, fc.INTEG_EULER, X[5:6], U[None, 5:65], dt)["traj"][0], g["thr_sim_euler"]) < 1e-14
    assert rel_err(fc.rollout(fc.MODEL_DI_THRUSTER_EULER, fc.INTEG_RK4, X[5:6], U[None, 5:65], dt)["traj"][0], g["thr_sim_rk4"]) < 1e-14
    for i, H in enumerate(g["H"]):
        assert abs(fc.window_rmse(fc.MODEL_DI_THRUSTER_EULER, fc.INTEG_EULER, X, U, int(H), dt) - g["thr_euler_rmse"][i]) < 1e-14
        assert abs(fc.window_rmse(fc.MODEL_DI_THRUSTER_EULER, fc.INTEG_RK4, X, U, int(H), dt) - g["thr_rk4_rmse"][i]) < 1e-14
    fc.set_di_gains(g["we_Klin"], g["we_Kang"])
    assert rel_err(fc.rollout(fc.MODEL_DI_WRENCH_EULER, fc.INTEG_EULER, X[5:6], TAU[None, 5:65], dt)["traj"][0], g["we_sim_euler"]) < 1e-14
    fc.set_di_gains(g["wq_Klin"], g["wq_Kang"])
    assert rel_err(fc.rollout(fc.MODEL_DI_WRENCH_QUAT, fc.INTEG_EULER, Xq[5:6], TAU[None, 5:65], dt)["traj"][0], g["wq_sim_euler"]) < 1e-14
    for i, H in enumerate(g["H"]):
        assert abs(fc.window_rmse(fc.MODEL_DI_WRENCH_QUAT, fc.INTEG_EULER, Xq, TAU, int(H), dt) - g["wq_euler_rmse"][i]) < 1e-14


def test_edmdc_lift_gram_fit_scores():
    g = load_golden("edmdc.npz")
    X, U, C = g["X"], g["U"], g["centers"]
    nt, gamma, ridge = int(g["n_train"]), float(g["gamma"]), float(g["ridge"])
    assert rel_err(ek.lift(X[:64], C, gamma), g["lift64"]) < 1e-13
    assert rel_err(ek.lift(X[7], C, gamma), g["lift1"]) < 1e-13
    assert rel_err(ek.rbf_mat(np.array([[0.3, -0.2, 1.0, 0.1, -0.2, 0.7, 0.4, -0.3, 0.2, 0.05, -0.1, 0.2]]),
                              np.array([[0.0] * 12, [0.1] * 12]), 3.0), g["rbf_kat"]) < 1e-15
    GtG, GtY, n = ek.gram([X[:nt]], [U[:nt]], C, gamma)
    assert n == nt - 1
    assert np.linalg.norm(GtG - g["GtG"]) / np.linalg.norm(g["GtG"]) < 1e-13
    assert np.linalg.norm(GtY - g["GtY"]) / np.linalg.norm(g["GtY"]) < 1e-13
    A, B = ek.solve_AB(GtG, GtY, ridge, 12 + C.shape[0])
    # fit() evaluates (pinv @ G.T) @ Y, fit_multi/our form pinv @ (G.T @ Y): equal to conditioning
    assert rel_err(A, g["A"]) < 1e-7 and rel_err(B, g["B"]) < 1e-7
    Xt, Ut = X[nt:], U[nt:]
    assert abs(ek.evaluate(Xt, Ut, C, gamma, A, B) - g["eval_rmse"]) < 1e-9
    for i, H in enumerate((1, 10, 100)):
        assert abs(ek.multistep_rmse(Xt, Ut, C, gamma, A, B, H) - g["ms_rmse"][i]) < 1e-8
        # with the reference's own A,B the restated propagation is exact to rounding
        assert abs(ek.multistep_rmse(Xt, Ut, C, gamma, g["A"], g["B"], H) - g["ms_rmse"][i]) < 1e-12
    assert rel_err(ek.simulate(Xt[0], Ut[:50], C, gamma, g["A"], g["B"]), g["sim50"]) < 1e-12


def test_edmdc_fit_multi_and_gamma3():
    g = load_golden("edmdc.npz")
    X, U = g["X"], g["U"]
    cuts = g["multi_cuts"]
    A, B = ek.fit([X[a:b] for a, b in cuts], [U[a:b] for a, b in cuts], g["multi_centers"], float(g["gamma"]), float(g["ridge"]))
    assert rel_err(A, g["multi_A"]) < 1e-9 and rel_err(B, g["multi_B"]) < 1e-9
    Xt, Ut = X[int(g["n_train"]):], U[int(g["n_train"]):]
    for i, H in enumerate((1, 10, 100)):
        assert abs(ek.multistep_rmse(Xt, Ut, g["multi_centers"], float(g["gamma"]), A, B, H) - g["multi_ms_rmse"][i]) < 1e-9
    nt = int(g["n_train"])
    A3, B3 = ek.fit([X[:nt]], [U[:nt]], g["g3_centers"], 3.0, 1e-1)
    assert rel_err(A3, g["g3_A"]) < 1e-8
    for i, H in enumerate((1, 10, 100)):
        assert abs(ek.multistep_rmse(Xt, Ut, g["g3_centers"], 3.0, A3, B3, H) - g["g3_ms_rmse"][i]) < 1e-9


def test_controls_stream_properties():
    u = controls.controls_iid(0x5EED, 3, 2, 5000, t0=100, nt=7)
    full = controls.controls_iid(0x5EED, 0, 5, 5000)
    assert np.array_equal(u, full[3:5, 100:107])
    assert full.min() >= -1.0 and full.max() < 1.0 and abs(full.mean()) < 5e-3
    a = controls.controls_ar1(1, 0, 2, 300)
    assert np.all(np.abs(a) <= 1.0) and np.abs(np.diff(a, axis=1)).max() < 0.15


def test_simscript_dataset_and_scores():
    """training/train_sim_brov2_koopmanEDMDc.py's data loop (numpy global RNG, seed 42) and its scores: the oracle
    rollout fed with the same draws reproduces the reference's states; the NumPy EDMDc restatement its scores."""
    g = load_golden("simscript.npz")
    N, dt, k = int(g["N"]), float(g["dt"]), int(g["k"])
    z = np.random.RandomState(42).randn(N, 20)
    U = np.empty((N, 8))
    u = np.zeros(8)
    for i in range(N):
        u = np.clip(0.98 * u + 0.02 * z[i, :8], -1.0, 1.0)
        U[i] = u
    assert np.array_equal(U, g["U"])
    r = fc.rollout(fc.MODEL_THRUSTER_EULER, fc.INTEG_EULER, np.zeros((1, 12)), U[None], dt)
    Xt = r["traj"][0, 1:]
    assert rel_err(Xt, g["X_true"]) < 1e-11
    X = Xt + z[:, 8:] * np.array([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3)
    assert np.max(np.abs(X - g["X"])) < 1e-12
    split = int(0.8 * N)
    A, B = ek.fit([g["X"][:split]], [g["U"][:split]], g["centers"], 1.0, 1e-3)
    assert rel_err(A, g["A"]) < 1e-7 and rel_err(B, g["B"]) < 1e-7
    Xte, Ute = g["X"][split - 1:], g["U"][split - 1:]
    got = [ek.evaluate(Xte, Ute, g["centers"], 1.0, A, B)] + [ek.multistep_rmse(Xte, Ute, g["centers"], 1.0, A, B, H) for H in (10, 100)]
    assert np.max(np.abs(np.array(got) - g["rmse"])) < 1e-8


def test_scalar_reference_shaped_restatement_matches_reference_rollouts():
    """oracle/fossen_scalar.py (the per-call NumPy loop bench.py times as `cpu_baseline_reference_shape`) against the
    reference's own states: config-2 stream, trajectory 0, first 400 RK4 and Euler steps, every 50th state."""
    import warnings
    from oracle import fossen_scalar as fs
    g = load_golden("fossen_rollouts.npz")
    T, dt, sub = 400, float(g["cfg2_dt"]), int(g["cfg2_sub"])
    U = controls.controls_iid(int(g["cfg2_seed"]), 0, 1, int(g["cfg2_T"]))[0][:T]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for integ, key in (("rk4", "cfg2_rk4"), ("euler", "cfg2_euler")):
            tr = fs.simulate(g["cfg2_x0"], U, dt, integ)
            assert rel_err(tr[::sub], g[key][0][: T // sub + 1]) < 1e-13, integ
        # and against the C oracle on a random state / command (one dynamics() call, fresh lag)
        rng = np.random.default_rng(5)
        x, u = rng.uniform(-0.5, 0.5, 12), rng.uniform(-1, 1, 8)
        xd = fs.ScalarBlueROV2().dynamics(x, u, 0.02)
    xo, _ = fc.rhs(0, x[None], u[None], 0.02, lag=np.zeros((1, 8, 3)))
    assert rel_err(xd, xo[0]) < 1e-12


def test_edmdc_fit_order_fixture_defaults_and_tank_settings():
    """KoopmanEDMDc.fit's own association (pinv G^T) Y (Koopman/koopmanEDMDc.py:97) at the class defaults and at the tank
    script's settings: the NumPy oracle in that order reproduces the reference's A, B and H = 1/10/100 RMSE; and the fixture
    itself records what fit_multi's order would have cost (1e-6 at H = 100 in the ill-conditioned 1 600-sample case)."""
    g = load_golden("edmdc_fit.npz")
    X, U, ntr = g["X"], g["U"], int(g["n_train"])
    for tag in ("def", "tank"):
        C, gamma, ridge = g[f"{tag}_centers"], float(g[f"{tag}_gamma"]), float(g[f"{tag}_ridge"])
        A, B = ek.fit_single(X[:ntr], U[:ntr], C, gamma, ridge)
        assert rel_err(A[:32, :32], g[f"{tag}_A_block"]) < 1e-7 and rel_err(B[:32], g[f"{tag}_B_block"]) < 1e-7
        assert abs(np.linalg.norm(A) / float(g[f"{tag}_A_fro"]) - 1) < 1e-9
        ms = np.array([ek.multistep_rmse(X[ntr:], U[ntr:], C, gamma, A, B, H) for H in (1, 10, 100)])
        assert np.max(np.abs(ms - g[f"{tag}_ms_rmse"])) < 1e-8, (tag, ms - g[f"{tag}_ms_rmse"])
        assert np.max(np.abs(g[f"{tag}_multi_order_ms_rmse"] - g[f"{tag}_ms_rmse"])) < 1e-8      # well conditioned: orders agree
    e = load_golden("edmdc.npz")
    Xs, Us, ns = e["X"], e["U"], int(e["n_train"])
    A, B = ek.fit_single(Xs[:ns], Us[:ns], g["small_centers"], 1.0, 1e-8)
    ms = np.array([ek.multistep_rmse(Xs[ns:], Us[ns:], g["small_centers"], 1.0, A, B, H) for H in (1, 10, 100)])
    assert np.max(np.abs(ms - g["small_ms_rmse"])) < 1e-7, ms - g["small_ms_rmse"]
    # the reference's own two orders, as recorded: 1.06e-6 apart at H = 100 -- above north_star's 1e-6
    assert abs(g["small_multi_order_ms_rmse"][2] - g["small_ms_rmse"][2]) > 5e-7


def test_kmeans_oracle_restates_sklearns_empty_cluster_rules():
    """oracle/kmeans_numpy.py against the fixture scikit-learn 1.7.2 produced (tools/gen_golden.py: gen_kmeans_empty) and against
    scikit-learn itself: relocation of an empty cluster to the farthest sample, no relocation when all distances are zero, the
    in-place averaging loop; and the fixed-point stand-in of the device loop (integer member sums) against the floating-point one."""
    import warnings
    from oracle import kmeans_numpy as kn
    g = load_golden("kmeans_empty.npz")
    X, C0 = g["a_X"], g["a_C0"]
    mean = X.mean(0)
    tol_abs = 1e-4 * np.mean(np.var(X, axis=0))
    C, lab, inertia, n_iter, nrel = kn.lloyd(X - mean, C0 - mean, 300, tol_abs)
    assert n_iter == int(g["a_n_iter"]) and nrel == 1 and np.max(np.abs(C + mean - g["a_centers"])) < 1e-12
    assert abs(inertia - float(g["a_inertia"])) <= 1e-10 * float(g["a_inertia"])
    Cf, labf, itf, nrf = kn.lloyd_fixed_point(X - mean, C0 - mean, 300, tol_abs)
    assert itf == n_iter and nrf == 1 and np.array_equal(labf, lab) and np.max(np.abs(Cf - C)) < 1e-13
    Xq, C0q = g["c_X"], g["c_C0"]
    Cq, _, _, itq, nrq = kn.lloyd(Xq, C0q, 300, 0.0)
    assert itq == int(g["c_n_iter"]) and nrq == 0 and np.array_equal(Cq, g["c_centers"]) and np.array_equal(Cq[1], [0.0, 0.0, 16.0, 0.0])
    assert np.array_equal(kn.lloyd_fixed_point(Xq, C0q, 300, 0.0)[0], g["c_centers"])
    # live against the installed scikit-learn (the pinned 1.7.2): several empties at once (np.argpartition's order)
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(4)
    X = np.cumsum(rng.normal(0, 0.05, (2500, 12)), 0)
    C0 = X[rng.choice(len(X), 48, replace=False)].copy()
    C0[3] += 60.0; C0[20] -= 45.0; C0[47] += 80.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=48, init=C0, n_init=1).fit(X)
    mean = X.mean(0)
    C, lab, _, n_iter, nrel = kn.lloyd(X - mean, C0 - mean, 300, 1e-4 * np.mean(np.var(X, axis=0)))
    assert n_iter == km.n_iter_ and nrel >= 1 and np.array_equal(lab, km.labels_) and np.max(np.abs(C + mean - km.cluster_centers_)) < 1e-12


def test_distance_bounds_never_hide_a_label_change():
    """The rule kmeans_bounds_kernel applies between two E-steps (csrc/kmeans.hip, round 4; restated in oracle/kmeans_numpy.py:
    bounds_step): starting from exact bounds, moved through ten M-steps WITHOUT re-evaluation, the upper bound stays above the true
    distance to the sample's centre and the lower bound below the true distance to every other centre -- so a sample the rule
    would skip (ub < lb) has the label the full scan gives.  Random data, several seeds; the samples the rule skips are counted so
    that the test cannot pass vacuously."""
    from oracle import kmeans_numpy as kn
    skipped = 0
    for seed in range(4):
        rng = np.random.default_rng(seed)
        n, k, N = (5, 24, 4000) if seed % 2 == 0 else (12, 40, 6000)
        X = np.cumsum(rng.normal(0, 0.05, (N, n)), 0)
        X -= X.mean(0)
        C = X[rng.choice(N, k, replace=False)].copy()
        for _ in range(4):                              # a few iterations first: later shifts are of a realistic size
            C, _ = kn.m_step(X, C, kn.e_step(X, C))
        lab = kn.e_step(X, C)
        D = np.sqrt(((X[:, None, :] - C[None, :, :]) ** 2).sum(-1))
        ub = D[np.arange(N), lab].copy()
        Dm = D.copy(); Dm[np.arange(N), lab] = np.inf
        lb = Dm.min(axis=1)
        labels0 = lab.copy()                            # the bounds refer to THESE labels throughout
        for it in range(10):
            C_new, _ = kn.m_step(X, C, kn.e_step(X, C))
            ub, lb = kn.bounds_step(ub, lb, labels0, C, C_new)
            C = C_new
            D = np.sqrt(((X[:, None, :] - C[None, :, :]) ** 2).sum(-1))
            own = D[np.arange(N), labels0]
            Dm = D.copy(); Dm[np.arange(N), labels0] = np.inf
            assert np.all(ub >= own - 1e-12), (seed, it, float(np.max(own - ub)))
            assert np.all(lb <= Dm.min(axis=1) + 1e-12), (seed, it, float(np.max(lb - Dm.min(axis=1))))
            safe = ub < lb
            assert np.array_equal(kn.e_step(X, C)[safe], labels0[safe]), (seed, it)
            skipped += int(safe.sum())
    assert skipped > 10000


def test_far_row_selection_is_numpys_own_introselect():
    """The rows `_relocate_empty_clusters_dense` moves empty clusters to are np.argpartition(distances, -n_empty)[:-n_empty-1:-1].
    tests/golden/farselect.npz holds NumPy's own answers with its SIMD dispatch DISABLED (tools/gen_farselect_golden.py), i.e. of its
    introselect: the oracle's restatement and the library's (edmdc_far_select_numpy, host only: no GPU needed) must both reproduce every
    case -- uniform values, many ties, NaNs, sorted / reversed / constant input, and Musser's median-of-3 killer, which takes the
    median-of-medians branch.  The fixture also records what THIS build host's NumPy returns with its dispatch on: it differs in half of
    the cases (x86-simd-sort), which is why the rule is pinned to the algorithm and the Python layer keeps a callback for the host's own."""
    import ctypes
    from bluerov2_dynamics_amd import _lib
    from oracle import kmeans_numpy as kn
    lib = _lib.load_library()
    g = load_golden("farselect.npz")
    names = sorted(k[:-2] for k in g.files if k.endswith("_d"))
    assert len(names) >= 20
    used_mom = 0
    for nm in names:
        d, ne, want = np.ascontiguousarray(g[nm + "_d"]), int(g[nm + "_n"]), g[nm + "_far"]
        sel = kn._Introselect(d)
        sel.select(0, len(d), len(d) - ne)
        used_mom += int(sel.used_median_of_medians)
        assert np.array_equal(np.array(sel.t[::-1][:ne]), want), nm
        out = np.empty(ne, dtype=np.int64)
        assert lib.edmdc_far_select_numpy(d.ctypes.data, len(d), ne, out.ctypes.data) == 0
        assert np.array_equal(out, want), nm
        # whatever the order, the rows are the n_empty largest (NaN = largest)
        key = np.where(np.isnan(d), np.inf, d)
        assert np.sort(key[out])[0] >= np.sort(key)[-ne] 
    assert used_mom >= 2                                          # the killer cases reach the median of medians of 5
    assert any(not np.array_equal(g[nm + "_far"], g[nm + "_far_simd"]) for nm in names) or len(g["simd_host_features"]) == 0
    assert lib.edmdc_far_select_numpy(None, 5, 1, None) == -1
