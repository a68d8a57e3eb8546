"""Host-side logic of the drop-in modules that needs no GPU: quaternion / rotation helpers, the
stand-alone ThrusterLag filter, the ridge solve, argument checking, sharding arithmetic."""
import numpy as np
import pytest

import os

from conftest import REPO, load_golden, rel_err


def test_quaternion_helpers_match_reference_fixture():
    from bluerov2_dynamics_amd.fossen import BlueROV2_wrench as qw
    g = load_golden("fossen_rhs_kat.npz")
    ang, Q = g["q_euler_in"], g["q_from_euler"]
    for i in range(len(ang)):
        assert rel_err(qw.euler_to_quat(*ang[i]), Q[i]) < 1e-15
        assert rel_err(np.array(qw.quat_to_euler(Q[i])), g["q_to_euler"][i]) < 1e-14
        assert abs(qw.quat_to_yaw(Q[i]) - g["q_to_yaw"][i]) < 1e-14
        assert rel_err(qw.quat_to_rotation_matrix(Q[i]), g["q_to_R"][i]) < 1e-15
        assert rel_err(qw.quat_multiply(Q[i], Q[(i + 1) % len(Q)]), g["q_mul"][i]) < 1e-15
        assert rel_err(qw.quat_derivative(Q[i], ang[i]), g["q_deriv"][i]) < 1e-15
    assert np.array_equal(qw.quat_normalize([0, 0, 0, 0]), [1, 0, 0, 0])
    assert np.array_equal(qw.quat_normalize([1e-13, 0, 0, 0]), [1, 0, 0, 0])
    with pytest.raises(ValueError):
        qw.quat_normalize([1, 0, 0])


def test_rotation_and_kinematics_helpers():
    from bluerov2_dynamics_amd.fossen.BlueROV2 import euler_kinematics_matrix, rotation_matrix
    from bluerov2_dynamics_amd.fossen.BlueROV2_wrench import euler_to_quat, quat_to_rotation_matrix
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.uniform(-1.4, 1.4, 3)
        R = rotation_matrix(*a)
        assert rel_err(R @ R.T, np.eye(3)) < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14
        assert rel_err(R, quat_to_rotation_matrix(euler_to_quat(*a))) < 1e-14
    J = euler_kinematics_matrix(0.3, np.pi / 2)        # clamp path: cos(theta) -> 1e-7
    assert J[0, 1] == pytest.approx(np.sin(0.3) * 1e7, rel=1e-9)


def test_thruster_lag_view_standalone_step():
    """ThrusterLag.step on the host = first lag sample of the reference fixture (x <- Bd F, y = Cc x)."""
    from bluerov2_dynamics_amd.fossen.BlueROV2 import ThrusterLag
    g = load_golden("fossen_rhs_kat.npz")
    U, dt = g["thr_U"], float(g["thr_dt"])
    store = np.zeros((8, 3))
    lags = [ThrusterLag(store, i) for i in range(8)]
    V = U[7]
    F = -140.3 * V**9 + 389.9 * V**7 - 404.1 * V**5 + 176.0 * V**3 + 8.9 * V
    y = [lags[i].step(F[i], dt) for i in range(8)]
    assert rel_err(store, g["thr_LAG"][0, 7]) < 1e-13
    assert rel_err(np.array(y), store @ np.array([0.0, 5.992, 3.317])) < 1e-15
    lags[2]._x = [1.0, 2.0, 3.0]
    assert np.array_equal(store[2], [1.0, 2.0, 3.0])


def test_ridge_solve_matches_reference_fixture():
    from bluerov2_dynamics_amd import engine
    g = load_golden("edmdc.npz")
    d = 12 + int(g["k"])
    A, B = engine.solve_AB(g["GtG"], g["GtY"], float(g["ridge"]), d)
    assert A.shape == (d, d) and B.shape == (d, 8) and A.flags["C_CONTIGUOUS"]
    assert rel_err(A, g["A"]) < 1e-7 and rel_err(B, g["B"]) < 1e-7


def test_shard_range_is_a_partition():
    from bluerov2_dynamics_amd.dist import shard_range
    for total in (0, 1, 7, 65536, 2**20, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_load_dataset_matches_reference_loader(tmp_path):
    """The CSV fixture has shuffled rows, duplicate time stamps and an inf: the loader must return exactly what the
    reference's load_dataset returned for it (fixture), and raise like the reference on missing columns."""
    import os
    import pandas as pd
    from bluerov2_dynamics_amd.data import load_dataset, write_dataset
    from conftest import GOLDEN
    g = load_golden("cfg5.npz")
    X, U, dt = load_dataset(os.path.join(GOLDEN, "cfg5_dataset.csv.gz"), verbose=False)
    assert np.array_equal(X, g["X"]) and np.array_equal(U, g["U"]) and dt == float(g["dt"])
    p = tmp_path / "d.csv"
    write_dataset(p, np.arange(5) * 0.02, X[:5], U[:5, :3], input_cols=["u1", "u2", "u3"])
    X2, U2, dt2 = load_dataset(p, verbose=False)                       # missing u4..u8 are zero-filled
    assert U2.shape == (5, 8) and np.all(U2[:, 3:] == 0.0) and abs(dt2 - 0.02) < 1e-12
    df = pd.read_csv(p)
    df.drop(columns=["theta"]).to_csv(p, index=False)
    with pytest.raises(ValueError, match="Missing state column"):
        load_dataset(p, verbose=False)
    df.drop(columns=["t"]).to_csv(p, index=False)
    with pytest.raises(ValueError, match="'t' time column"):
        load_dataset(p, verbose=False)


def test_load_dataset_wrench_and_quaternion_variants():
    """Wrench-input and quaternion-state loaders == the reference's (train_tank_brov2_wrench_comp.py:172-200,
    train_tank_brov2_wrench_quat.py:180-245) on the same file, incl. the Euler -> quaternion conversion path."""
    import os
    from bluerov2_dynamics_amd.data import load_dataset
    from conftest import GOLDEN
    g = load_golden("cfg5w.npz")
    path = os.path.join(GOLDEN, "cfg5w_dataset.csv.gz")
    X, U, dt = load_dataset(path, verbose=False, variant="wrench")
    assert np.array_equal(X, g["we_X"]) and np.array_equal(U, g["we_U"]) and dt == float(g["we_dt"])
    Xq, Uq, dtq = load_dataset(path, verbose=False, variant="quat")
    assert Xq.shape[1] == 13 and np.max(np.abs(Xq - g["wq_X"])) < 1e-15 and np.array_equal(Uq, g["wq_U"]) and dtq == float(g["wq_dt"])
    assert np.max(np.abs(np.linalg.norm(Xq[:, 3:7], axis=1) - 1.0)) < 1e-15
    with pytest.raises(ValueError):
        load_dataset(path, verbose=False, variant="nope")


def test_bluerov_torch_rhs_matches_reference_fixture():
    """The PINc physics-loss model stays on PyTorch (last row): same numbers as fossen/bluerov_torch.py."""
    import torch
    from bluerov2_dynamics_amd.fossen.bluerov_torch import bluerov_compute, ssa
    g = load_golden("torch_rhs.npz")
    x, u = torch.from_numpy(g["x"]), torch.from_numpy(g["u"])
    assert rel_err(bluerov_compute(0.0, x, u).numpy(), g["xdot64"]) < 1e-14
    assert rel_err(bluerov_compute(0.0, x.float(), u.float()).numpy(), g["xdot32"]) < 1e-5
    one = bluerov_compute(0.0, x[3], u[3])
    assert one.shape == (1, 9) and rel_err(one.numpy(), g["xdot_1d"]) < 1e-14
    assert np.max(np.abs(ssa(torch.from_numpy(g["ang"])).numpy() - g["ssa"])) < 1e-14


def test_driver_entry_points_exist_and_bench_parses_its_flags():
    """__graft_entry__ exposes build() and smoke(); bench.py accepts the driver's flags (checked without a GPU via --help)."""
    import importlib.util
    import os
    import subprocess
    import sys
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(REPO, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    assert callable(ge.build) and callable(ge.smoke)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_thruster_geometry_entries_are_views_of_one_block():
    """thrusters_r[i]["r" | "dir"] of the drop-in vehicle (fossen/BlueROV2.py:172-232 holds plain dicts of arrays): edits in
    place, assignment of a new array to a key, and replacement of a whole entry all land in the one [8,2,3] block whose bytes
    the per-call path compares -- no GPU needed to check the container classes."""
    from bluerov2_dynamics_amd.fossen.BlueROV2 import _ThrusterEntry, _ThrusterList
    geom = np.arange(48, dtype=float).reshape(8, 2, 3)
    lst = _ThrusterList(_ThrusterEntry(geom[i]) for i in range(8))
    before = geom.tobytes()
    lst[2]["r"][1] += 0.5                                         # in place
    assert geom[2, 0, 1] == 13.5 + 0.0 and geom.tobytes() != before
    lst[3]["dir"] = [9.0, 8.0, 7.0]                               # key assignment copies into the view
    assert np.array_equal(geom[3, 1], [9.0, 8.0, 7.0]) and lst[3]["dir"].base is not None
    lst[0] = {"r": np.ones(3), "dir": np.zeros(3)}                # entry replacement copies too
    assert np.array_equal(geom[0], [[1, 1, 1], [0, 0, 0]]) and isinstance(lst[0], _ThrusterEntry)
    lst[1]["label"] = "front-left"                                # other keys behave like a dict
    assert lst[1]["label"] == "front-left" and set(lst[1]) == {"r", "dir", "label"}
    with np.testing.assert_raises(ValueError):
        lst[4]["r"] = np.zeros(4)
    with np.testing.assert_raises(TypeError):
        lst[0:2] = []
    # round-2 advice: dict.update / setdefault / |= bypass __setitem__ in CPython -- they must land in the block too
    lst[5].update(r=[0.1, 0.2, 0.3])
    assert np.array_equal(geom[5, 0], [0.1, 0.2, 0.3]) and np.shares_memory(lst[5]["r"], geom)
    lst[5].update({"dir": np.array([0.0, 0.0, -1.0])}, note="x")
    assert np.array_equal(geom[5, 1], [0.0, 0.0, -1.0]) and lst[5]["note"] == "x" and np.shares_memory(lst[5]["dir"], geom)
    lst[6] |= {"r": [3.0, 2.0, 1.0]}
    assert np.array_equal(geom[6, 0], [3.0, 2.0, 1.0]) and isinstance(lst[6], _ThrusterEntry)
    assert lst[6].setdefault("r", [0, 0, 0]) is lst[6]["r"] and np.array_equal(geom[6, 0], [3.0, 2.0, 1.0])
    assert lst[6].setdefault("tag", 7) == 7
    for bad in (lambda: lst[6].pop("r"), lambda: lst[6].clear(), lambda: lst[6].__delitem__("dir"), lambda: lst[6].popitem()):
        with np.testing.assert_raises(KeyError):
            bad()
    assert lst[6].pop("tag") == 7


def test_bag_table_native_helper_equals_the_python_walk():
    """engine.BagTable (the bookkeeping of fit_multi's trajectory lists for brov_upload_bags) through the CPython helper
    (csrc/bagtable.c, buffer protocol) and through its Python loop: the same lengths, offsets, input rows and host addresses on a
    list with empty bags, a Fortran-ordered, a float32 and a nested-list bag, inputs one row short; the same errors."""
    import ctypes
    from bluerov2_dynamics_amd import _build, engine
    _build.build_bagtable()
    if engine._bagtable is None:                       # (the helper was built after the package had been imported)
        import importlib
        engine._bagtable = importlib.import_module("bluerov2_dynamics_amd._bagtable")
    rng = np.random.default_rng(0)
    Xs = [rng.normal(size=(int(L), 12)) for L in rng.integers(0, 50, 300)]
    Us = [rng.normal(size=(len(x), 8)) for x in Xs]
    Xs[5] = np.asfortranarray(Xs[5])
    Xs[9] = Xs[9].astype(np.float32)
    Us[17] = Us[17][:max(len(Us[17]) - 1, 0)]
    Xs[20] = Xs[20].tolist() if len(Xs[20]) else Xs[20]
    a = engine.BagTable(Xs, Us, 12, 8)
    native, engine._bagtable = engine._bagtable, None
    try:
        b = engine.BagTable(Xs, Us, 12, 8)
    finally:
        engine._bagtable = native
    for key in ("lens", "u_rows", "offsets"):
        assert np.array_equal(getattr(a, key), getattr(b, key)), key
    assert a.rows == b.rows == sum(len(x) for x in Xs) and a.pairs == b.pairs
    for t in (a, b):
        for i in range(len(Xs)):
            if t.lens[i]:
                got = np.ctypeslib.as_array((ctypes.c_double * int(t.lens[i] * 12)).from_address(int(t.x_ptrs[i]))).reshape(-1, 12)
                assert np.array_equal(got, np.asarray(Xs[i], dtype=float).reshape(-1, 12)), i
    for maker in (lambda: engine.BagTable([np.zeros((3, 11))], [np.zeros((3, 8))], 12, 8),):
        with pytest.raises(AssertionError):
            maker()
    with pytest.raises(ValueError):
        engine.BagTable([np.zeros((5, 12))], [np.zeros((3, 8))], 12, 8)


def _parse_like_the_driver(stdout):
    """what the driver does with bench.py's stdout: the last non-empty line is ONE JSON object"""
    import json
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) <= 4096, len(lines[0])
    return json.loads(lines[0])


@pytest.mark.parametrize("record", ["r05_bench.json", "r06_bench_details.json"])
def test_bench_compact_line_of_a_recorded_run_fits_the_driver(record):
    """bench.compact_line on the full record of a real run (profiles/): <= 4096 bytes with the contract fields, `roofline`, `cpu_baseline`
    and a flat summary -- also when eight ranks report (per-rank times in the summary) -- and it parses as the driver parses it."""
    import json
    import bench
    path = os.path.join(REPO, "profiles", record)
    if not os.path.exists(path):
        pytest.skip(f"{record} not recorded yet")
    full = json.load(open(path))
    full.setdefault("cpu_seconds_budget_per_leg", 3.0)
    for ranks in (1, 8):
        d = json.loads(json.dumps(full))
        d["n_gpus"] = ranks
        if "config4" in d:
            d["config4"]["per_rank_ms"] = [7126.156140999228 / ranks + i for i in range(ranks)]
            d["config4"]["rccl_ranks"] = ranks
        c = _parse_like_the_driver(json.dumps(bench.compact_line(d, "bench_details.json"), separators=(",", ":")) + "\n")
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline", "cpu_baseline", "verified", "summary", "details"):
            assert k in c, k
        assert c["n_gpus"] == ranks and c["metric"] == "rk4_rollout_steps_per_s" and c["dtype"] == "f64" and "workload" in c["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in c["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c["cpu_baseline"], k
        assert abs(c["value"] - full["value"]) / full["value"] < 1e-9 and abs(c["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-9
        assert all(not isinstance(v, dict) for v in c["summary"].values())
        assert len(c["summary"].get("cfg4_per_rank_ms", [0] * ranks)) == ranks
    # a record too long for the line loses summary entries, never a contract field
    d = json.loads(json.dumps(full))
    d["config"]["workload"] = d["config"]["workload"] + " x" * 3000
    c = bench.compact_line(d, "bench_details.json")
    assert len(json.dumps(c, separators=(",", ":"))) <= 4096 and "roofline" in c and "cpu_baseline" in c


def test_default_solve_is_numpy_pinv_wherever_the_route_matters():
    """engine._host_pinv_route("auto") -- the class default -- on the NumPy oracle's Gram of the ill-conditioned fixture (class-default ridge
    1e-8, gamma 0.05: smallest eigenvalue 2.7e-12 of the largest): bit for bit numpy.linalg.pinv, the reference's call
    (Koopman/koopmanEDMDc.py:97), and the reference's scores follow; at gamma 0.2 (2.5e-9, kappa_1 bound 2.2e9) the Cholesky inverse is
    taken and gives the same scores to 1e-8; on the tank settings (ridge 0.1) likewise to 1e-9.  A matrix between the two thresholds takes
    the eigendecomposition.  The unconditional "eigh" route at gamma 0.05 is measurably off (that was round 5's default)."""
    from oracle import edmdc_numpy as ek
    from bluerov2_dynamics_amd import engine
    e, z = load_golden("edmdc_fit.npz"), load_golden("edmdc_illcond.npz")
    X, U, ntr = e["X"], e["U"], int(e["n_train"])
    for tag, want, tol in (("g005", "pinv", 1e-9), ("g02", "cholesky", 1e-8)):
        gamma, C, ridge = float(z[f"{tag}_gamma"]), z[f"{tag}_centers"], 1e-8
        Z, Zp = ek.lift(X[:ntr - 1], C, gamma), ek.lift(X[1:ntr], C, gamma)
        G = np.hstack([Z, U[:ntr - 1]])
        GtG = G.T @ G
        Pa, route = engine._host_pinv_route(GtG, ridge, "auto")
        assert route == want, (tag, route)
        if want == "pinv":
            assert np.array_equal(Pa, engine._host_pinv(GtG, ridge, "host"))
        else:
            assert np.array_equal(Pa, Pa.T)
        M = (Pa @ G.T @ Zp).T
        A, B = M[:, :Z.shape[1]], M[:, Z.shape[1]:]
        sc = np.array([ek.multistep_rmse(X[ntr:], U[ntr:], C, gamma, A, B, H) for H in (1, 10, 100)])
        assert np.max(np.abs(sc - z[f"{tag}_ms_rmse"]) / np.maximum(1.0, z[f"{tag}_ms_rmse"])) < tol, tag
    # between the thresholds: kappa_1 says "cannot tell", the eigenvalues say "safe"
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.normal(size=(60, 60)))
    S = (Q * np.geomspace(1.0, 2.5e-10, 60)) @ Q.T
    P, route = engine._host_pinv_route(0.5 * (S + S.T), 0.0, "auto")
    assert route in ("eigh", "cholesky") and np.linalg.norm(P @ S - np.eye(60)) < 1e-4
    assert engine._host_pinv_route(np.diag([1.0, 1e-13]), 0.0, "auto")[1] == "pinv"
    assert engine._host_pinv_route(-np.eye(3), 0.0, "auto")[1] == "pinv"                # not positive definite: the reference's route
    with pytest.raises(ValueError):
        engine._host_pinv(np.eye(3), 0.1, "svd")


def test_two_rank_rehearsal_line_parses_like_the_drivers_scaling_run():
    """profiles/r06_rehearsal_2ranks_shared_gpu.json = the stdout of `bench.py --gpus 2 --steps 20 --warmup 5` with two ranks sharing the one
    GPU of the development box over gloo (BROV2_BENCH_SHARE_GPU=1 BROV2_BENCH_BACKEND=gloo; RCCL refuses two ranks per device): the first
    real multi-GPU run must not be the first time that code path prints.  Parsed the way the driver parses: ONE line, <= 4 096 bytes."""
    path = os.path.join(REPO, "profiles", "r06_rehearsal_2ranks_shared_gpu.json")
    if not os.path.exists(path):
        pytest.skip("rehearsal not recorded yet")
    c = _parse_like_the_driver(open(path).read())
    assert c["n_gpus"] == 2 and c["steps"] == 20 and c["warmup"] == 5 and c["scaling"] == "weak" and c["metric"] == "rk4_rollout_steps_per_s"
    assert abs(c["value"] - 2 * 65536 * 5000 * 20 / (c["ms_per_step"] * 20e-3)) / c["value"] < 1e-6          # whole-job aggregate over both ranks
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    sm = c["summary"]
    assert sm["cfg4_rccl_ranks"] == 2 and len(sm["cfg4_per_rank_ms"]) == 2 and sm["cfg4_verified"] is True and sm["cfg4_total_rollouts"] == 1 << 20
    assert sm["fit_sharded_identical_on_all_ranks"] is True and sm["fit_sharded_samples_per_s"] > 0 and c["verified"]["ok"] is True
