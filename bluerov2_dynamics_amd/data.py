"""On-disk boundary: the 50 Hz CSV schema of the reference (rosbags/bag2csv.py:462-465:
`t,x,y,z,phi,theta,psi,u,v,w,p,q,r,u1..u8`, wrench variant `Fx,Fy,Fz,Mx,My,Mz`) and the cleaning rules of the
training scripts' load_dataset (training/train_tank_brov2_full_comparison.py:82-110).  Host-side glue (pandas)."""
import numpy as np

STATE_COLS = ["x", "y", "z", "phi", "theta", "psi", "u", "v", "w", "p", "q", "r"]
THRUSTER_COLS = [f"u{i}" for i in range(1, 9)]
WRENCH_COLS = ["Fx", "Fy", "Fz", "Mx", "My", "Mz"]


def load_dataset(csv_path, input_cols=None, verbose=True):
    """Returns (X [N,12], U [N,nu], dt).  Missing state column / missing `t` raise ValueError; missing input columns are
    zero-filled; rows are sorted by t, duplicate t dropped (first kept), +-inf -> NaN, rows with a NaN state dropped;
    dt = median(diff(t)) (0.05 for a single row) -- exactly the reference's behaviour."""
    import pandas as pd
    input_cols = list(THRUSTER_COLS if input_cols is None else input_cols)
    df = pd.read_csv(csv_path)
    for c in STATE_COLS:
        if c not in df.columns:
            raise ValueError(f"Missing state column: {c}")
    for c in input_cols:
        if c not in df.columns:
            df[c] = 0.0
    if "t" not in df.columns:
        raise ValueError("CSV must contain a 't' time column.")
    df = df.sort_values("t").drop_duplicates(subset="t")
    df = df.replace([np.inf, -np.inf], np.nan).dropna(subset=STATE_COLS)
    X = df[STATE_COLS].to_numpy(float)
    U = df[input_cols].to_numpy(float)
    t = df["t"].to_numpy(float)
    dt = float(np.median(np.diff(t))) if len(t) > 1 else 0.05
    if verbose:
        print(f"[i] Samples: {len(df)} | median dt = {dt:.5f}s (~{1.0 / max(dt, 1e-9):.2f} Hz)")
    return X, U, dt


def write_dataset(csv_path, t, X, U, input_cols=None):
    """Write a CSV in the reference's schema (used by the examples / tests to make synthetic recordings)."""
    import pandas as pd
    input_cols = list(THRUSTER_COLS if input_cols is None else input_cols)
    df = pd.DataFrame(np.column_stack([t, X, U]), columns=["t"] + STATE_COLS + input_cols)
    df.to_csv(csv_path, index=False, float_format="%.12g")
