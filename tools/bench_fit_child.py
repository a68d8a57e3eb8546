"""Child process of bench.py's `recorded_shape` leg: KoopmanEDMDc.fit() at the size the reference's own log quotes
(training/best_results.txt:761,798: N = 45 823 samples, n = 12, r = 8, 500 RBFs, gamma = 3, ridge = 0.1 -- the settings of
training/train_tank_brov2_full_comparison.py:40-44) on HOST arrays, as the FIRST thing a fresh process does, then again (warm).

    python tools/bench_fit_child.py gpu <data.npz> [pinv] [arrays]   the drop-in class (libbrov2.so); arrays = native (default: no torch in the
                                                              process unless $BROV2_TORCH=1) or torch (the torch-tensor path of rounds 1-5)
    python tools/bench_fit_child.py csv <recording.csv> <k> <gamma> <ridge> [warm_up]   what the reference's tank scripts do: parse the 50 Hz CSV
                                                              (data.load_dataset = their load_dataset), then fit -- optionally with
                                                              bluerov2_dynamics_amd.warm_up() right after the imports
    python tools/bench_fit_child.py cpu <data.npz>            the NumPy / scikit-learn restatement of the reference's fit() in its own
                                                              shape (oracle/edmdc_numpy.py; bench.py's cpu_baseline leg -- the parent
                                                              sets OMP_NUM_THREADS=4, the reference's import-time default,
                                                              Koopman/koopmanEDMDc.py:23-25)
Prints one JSON object on the last line of stdout."""
import json
import os
import sys
import time

T_START = time.perf_counter()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def csv_script(path, k, gamma, ridge, warm):
    """imports -> [warm_up()] -> load_dataset(csv) -> KoopmanEDMDc.fit: the shape of training/train_tank_brov2_koopmanEDMDc.py's start"""
    import bluerov2_dynamics_amd
    from bluerov2_dynamics_amd import _lib, data
    from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
    t_imp = time.perf_counter() - T_START
    if warm:
        bluerov2_dynamics_amd.warm_up()
    t0 = time.perf_counter()
    X, U, dt = data.load_dataset(path, verbose=False)
    t_load = time.perf_counter() - t0
    m = KoopmanEDMDc(state_dim=X.shape[1], input_dim=U.shape[1], n_rbfs=k, gamma=gamma, ridge=ridge)
    t0 = time.perf_counter()
    m.fit(X, U)
    t_fit = time.perf_counter() - t0
    import numpy as np
    print(json.dumps({"mode": "csv", "warm_up": bool(warm), "rows": int(X.shape[0]), "imports_s": t_imp, "load_dataset_s": t_load, "first_fit_s": t_fit,
                      "process_start_to_first_fit_done_s": time.perf_counter() - T_START, "torch_imported": "torch" in sys.modules,
                      "finite": bool(np.isfinite(m.A_).all()), "hip_runtime": _lib.hip_runtime}), flush=True)


def main():
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "csv":
        return csv_script(path, int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), len(sys.argv) > 6 and sys.argv[6] == "warm_up")
    import numpy as np
    z = np.load(path)
    out = {"mode": mode}
    cases = [("thruster_12_8", z["X"], z["U"])]
    if "Xq" in z.files:
        cases.append(("quaternion_13_6", z["Xq"], z["Uq"]))
    k, gamma, ridge = int(z["k"]), float(z["gamma"]), float(z["ridge"])
    Hs = (1, 10, 100)
    if mode == "gpu":
        import hashlib
        pinv = sys.argv[3] if len(sys.argv) > 3 else "auto"
        arrays = sys.argv[4] if len(sys.argv) > 4 else "native"
        out["data_loaded_s_since_start"] = time.perf_counter() - T_START
        t0 = time.perf_counter()
        from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc     # binds libbrov2.so lazily (first Context)
        out["import_s"] = time.perf_counter() - t0
        out["pinv"], out["arrays"] = pinv, arrays
        for name, X, U in cases:
            n, r = X.shape[1], U.shape[1]
            calls, first_done = [], None
            for rep in range(7 if name == cases[0][0] else 4):
                m = KoopmanEDMDc(state_dim=n, input_dim=r, n_rbfs=k, gamma=gamma, ridge=ridge, pinv=pinv, arrays=arrays)
                t0 = time.perf_counter()
                m.fit(X, U)
                calls.append(time.perf_counter() - t0)
                if rep == 0 and name == cases[0][0]:
                    first_done = time.perf_counter() - T_START        # interpreter start -> imports -> data -> first fit() returned
                    out["first_fit_AB_sha256"] = hashlib.sha256(m.A_.tobytes() + m.B_.tobytes() + m.centers_.tobytes()).hexdigest()[:16]
            t0 = time.perf_counter()
            scores = [m.multistep_rmse(X, U, H) for H in Hs]
            sc_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            scores = [m.multistep_rmse(X, U, H) for H in Hs]
            sc2_s = time.perf_counter() - t0
            out[name] = {"rows": int(X.shape[0]), "n": n, "r": r, "k": k, "fit_calls_s": calls,
                         "process_start_to_first_fit_done_s": first_done,
                         "multistep_rmse_H1_10_100": scores, "multistep_rmse_three_calls_s": [sc_s, sc2_s],
                         "finite": bool(np.isfinite(m.A_).all() and np.isfinite(m.B_).all()),
                         "checksum": float(np.abs(m.A_).sum() + np.abs(m.B_).sum())}
        from bluerov2_dynamics_amd import _lib
        out["torch_imported"] = "torch" in sys.modules
        out["hip_runtime"] = _lib.hip_runtime
        out["process_total_s"] = time.perf_counter() - T_START
    else:
        from sklearn.cluster import KMeans
        from oracle import edmdc_numpy as ek
        try:
            from threadpoolctl import threadpool_info
            out["threadpools"] = [{"api": t_["user_api"], "threads": t_["num_threads"]} for t_ in threadpool_info()]
        except Exception:
            out["threadpools"] = None
        out["OMP_NUM_THREADS"] = os.environ.get("OMP_NUM_THREADS")
        name, X, U = cases[0]
        calls, stages = [], None
        for rep in range(2):
            t0 = time.perf_counter()
            C = KMeans(n_clusters=k, n_init="auto", random_state=0).fit(X).cluster_centers_        # Koopman/koopmanEDMDc.py:85
            t1 = time.perf_counter()
            A, B = ek.fit_single(X, U, C, gamma, ridge)                                            # :88-101, the reference's own order
            t2 = time.perf_counter()
            calls.append(t2 - t0)
            stages = {"kmeans_s": t1 - t0, "lift_gram_pinv_products_s": t2 - t1}
        t0 = time.perf_counter()
        s10 = ek.multistep_rmse(X, U, C, gamma, A, B, 10)
        out[name] = {"rows": int(X.shape[0]), "fit_calls_s": calls, "stages_second_call": stages, "multistep_rmse_H10": s10,
                     "multistep_rmse_H10_s": time.perf_counter() - t0, "checksum": float(np.abs(A).sum() + np.abs(B).sum())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
