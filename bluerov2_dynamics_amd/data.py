"""On-disk boundary: the 50 Hz CSV schema of the reference (rosbags/bag2csv.py:462-465:
`t,x,y,z,phi,theta,psi,u,v,w,p,q,r,u1..u8`, wrench variant `Fx,Fy,Fz,Mx,My,Mz`) and the cleaning rules of the
training scripts' load_dataset (training/train_tank_brov2_full_comparison.py:82-110).  Host-side glue (pandas)."""
import numpy as np

STATE_COLS = ["x", "y", "z", "phi", "theta", "psi", "u", "v", "w", "p", "q", "r"]
QUAT_STATE_COLS = ["x", "y", "z", "qw", "qx", "qy", "qz", "u", "v", "w", "p", "q", "r"]
THRUSTER_COLS = [f"u{i}" for i in range(1, 9)]
WRENCH_COLS = ["Fx", "Fy", "Fz", "Mx", "My", "Mz"]


def load_dataset(csv_path, input_cols=None, verbose=True, variant="thruster"):
    """Returns (X [N,nx], U [N,nu], dt).  Missing state column / missing `t` raise ValueError; missing input columns are
    zero-filled; rows are sorted by t, duplicate t dropped (first kept), +-inf -> NaN, rows with a NaN state dropped;
    dt = median(diff(t)) (0.05 for a single row) -- exactly the reference's behaviour.
      variant "thruster": Euler-angle state, inputs u1..u8        (training/train_tank_brov2_full_comparison.py:82-110)
      variant "wrench":   Euler-angle state, inputs Fx..Mz        (training/train_tank_brov2_wrench_comp.py:172-200)
      variant "quat":     quaternion state [x y z qw qx qy qz u v w p q r], inputs Fx..Mz; a legacy Euler-angle file is
                          converted (Z-Y-X) and every quaternion normalised   (training/train_tank_brov2_wrench_quat.py:180-245)"""
    import pandas as pd
    if variant not in ("thruster", "wrench", "quat"):
        raise ValueError("variant must be 'thruster', 'wrench' or 'quat'")
    if input_cols is None:
        input_cols = THRUSTER_COLS if variant == "thruster" else WRENCH_COLS
    input_cols = list(input_cols)
    state_cols = QUAT_STATE_COLS if variant == "quat" else STATE_COLS
    df = pd.read_csv(csv_path)
    if variant == "quat" and all(c in df.columns for c in ("phi", "theta", "psi")) and not all(c in df.columns for c in ("qw", "qx", "qy", "qz")):
        if verbose:
            print("[warn] Euler angles detected in dataset; converting to quaternions...")
        phi, theta, psi = (df[c].to_numpy(dtype=float) for c in ("phi", "theta", "psi"))
        c1, s1 = np.cos(phi * 0.5), np.sin(phi * 0.5)
        c2, s2 = np.cos(theta * 0.5), np.sin(theta * 0.5)
        c3, s3 = np.cos(psi * 0.5), np.sin(psi * 0.5)
        q = np.vstack([c3 * c2 * c1 + s3 * s2 * s1, c3 * c2 * s1 - s3 * s2 * c1, c3 * s2 * c1 + s3 * c2 * s1, s3 * c2 * c1 - c3 * s2 * s1]).T
        q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
        df["qw"], df["qx"], df["qy"], df["qz"] = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    for c in state_cols:
        if c not in df.columns:
            raise ValueError(f"Missing state column: {c}")
    for c in input_cols:
        if c not in df.columns:
            df[c] = 0.0
    if "t" not in df.columns:
        raise ValueError("CSV must contain a 't' time column.")
    df = df.sort_values("t").drop_duplicates(subset="t")
    df = df.replace([np.inf, -np.inf], np.nan).dropna(subset=state_cols)
    if variant == "quat":
        q = df[["qw", "qx", "qy", "qz"]].to_numpy(dtype=float)
        q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
        df.loc[:, ["qw", "qx", "qy", "qz"]] = q
    X = df[state_cols].to_numpy(float)
    U = df[input_cols].to_numpy(float)
    t = df["t"].to_numpy(float)
    dt = float(np.median(np.diff(t))) if len(t) > 1 else 0.05
    if verbose:
        print(f"[i] Samples: {len(df)} | median dt = {dt:.5f}s (~{1.0 / max(dt, 1e-9):.2f} Hz)")
    return X, U, dt


def load_dataset_dev(csv_path, input_cols=None, verbose=True, variant="thruster", ctx=None):
    """load_dataset with the samples left in HBM: (Xd [N,nx], Ud [N,nu] as engine.DevArray, dt, (X, U) host copies).  The device arrays go
    straight into KoopmanEDMDc.fit / fit_multi (no second upload) and into the engine's `_dev` entry points (rollout_dev windows,
    gram_dev, multistep).  SURVEY 8(f)4: "a loader feeding device buffers directly" -- the parsing itself stays pandas (45 823 rows: 0.1 s,
    not on the hot path); what this saves is the staging of every later call."""
    from . import engine
    X, U, dt = load_dataset(csv_path, input_cols=input_cols, verbose=verbose, variant=variant)
    ctx = ctx or engine.default_context()
    return engine.DevArray.from_host(ctx, X), engine.DevArray.from_host(ctx, U), dt, (X, U)


def write_dataset(csv_path, t, X, U, input_cols=None):
    """Write a CSV in the reference's schema (used by the examples / tests to make synthetic recordings)."""
    import pandas as pd
    input_cols = list(THRUSTER_COLS if input_cols is None else input_cols)
    df = pd.DataFrame(np.column_stack([t, X, U]), columns=["t"] + STATE_COLS + input_cols)
    df.to_csv(csv_path, index=False, float_format="%.12g")
