"""NumPy restatement of Koopman/koopmanEDMDc.py.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Pinned by tests/golden/edmdc.npz (tests/test_oracle_golden.py).  Centres are an INPUT here:
the reference obtains them from sklearn KMeans(n_init="auto", random_state=0)
(Koopman/koopmanEDMDc.py:85,126; scikit-learn 1.7.2, uv.lock:1338), which both the oracle's
callers and the product call directly -- it is a third-party dependency, not restated.
"""
import numpy as np


def rbf_mat(X, C, gamma):
    """Koopman/koopmanEDMDc.py:41-48 -- the EXPANDED distance form, kept as is."""
    x2 = np.sum(X ** 2, axis=1)[:, None]
    c2 = np.sum(C ** 2, axis=1)[None, :]
    return np.exp(-gamma * (x2 + c2 - 2 * X @ C.T))


def lift(X, C, gamma):
    """_lift: Koopman/koopmanEDMDc.py:221-236 (phi(x) = [x, rbf(x)])."""
    X = np.asarray(X, dtype=float)
    if X.ndim == 1:
        return np.hstack([X, rbf_mat(X[None, :], C, gamma).ravel()])
    if X.ndim == 2:
        return np.hstack([X, rbf_mat(X, C, gamma)])
    raise ValueError("x must have ndim 1 or 2")


def gram(X_list, U_list, C, gamma):
    """G^T G and G^T Y over bags without cross-bag pairs (fit_multi :129-147; fit :89-97 is
    the single-bag case).  Returns (GtG [p,p], GtY [p,d], n_pairs)."""
    d = X_list[0].shape[1] + C.shape[0]
    p = d + U_list[0].shape[1]
    GtG = np.zeros((p, p))
    GtY = np.zeros((p, d))
    n = 0
    for X, U in zip(X_list, U_list):
        if len(X) < 2:
            continue
        Z = lift(X[:-1], C, gamma)
        Zp = lift(X[1:], C, gamma)
        G = np.hstack([Z, U[:-1]])
        GtG += G.T @ G
        GtY += G.T @ Zp
        n += len(X) - 1
    return GtG, GtY, n


def solve_AB(GtG, GtY, ridge, d):
    """M = pinv(G^T G + ridge I) (G^T Y); A = M^T[:, :d], B = M^T[:, d:]  (fit_multi :147-151)."""
    M = np.linalg.pinv(GtG + ridge * np.eye(GtG.shape[0])) @ GtY
    M = M.T
    return M[:, :d], M[:, d:]


def fit_single(X, U, C, gamma, ridge):
    """KoopmanEDMDc.fit's own association (Koopman/koopmanEDMDc.py:94-101): M = pinv(G^T G + ridge I) @ G.T @ Y evaluated
    left to right, i.e. (P G^T) Y -- better conditioned than fit_multi's P (G^T Y) (:147)."""
    Z = lift(X[:-1], C, gamma)
    Zp = lift(X[1:], C, gamma)
    G = np.hstack([Z, U[:-1]])
    M = np.linalg.pinv(G.T @ G + ridge * np.eye(G.shape[1])) @ G.T @ Zp
    M = M.T
    d = Z.shape[1]
    return M[:, :d], M[:, d:]


def fit(X_list, U_list, C, gamma, ridge):
    GtG, GtY, _ = gram(X_list, U_list, C, gamma)
    d = X_list[0].shape[1] + C.shape[0]
    return solve_AB(GtG, GtY, ridge, d)


def evaluate(X, U, C, gamma, A, B):
    """Koopman/koopmanEDMDc.py:157-170."""
    n = X.shape[1]
    Z = lift(X[:-1], C, gamma)
    Zh = Z @ A.T + U[:-1] @ B.T
    return float(np.sqrt(np.mean((X[1:] - Zh[:, :n]) ** 2)))


def multistep_rmse(X, U, C, gamma, A, B, H):
    """Koopman/koopmanEDMDc.py:172-200."""
    N, n = X.shape
    ns = N - H
    Z = lift(X[:ns], C, gamma)
    for t in range(H):
        Z = Z @ A.T + U[t:t + ns] @ B.T
    return float(np.sqrt(np.mean((X[H:] - Z[:, :n]) ** 2)))


def simulate(x0, U_seq, C, gamma, A, B):
    """Koopman/koopmanEDMDc.py:202-216."""
    n = len(x0)
    out = np.zeros((len(U_seq) + 1, n))
    out[0] = x0
    z = lift(np.asarray(x0, float), C, gamma)
    for t, u in enumerate(U_seq):
        z = A @ z + B @ u
        out[t + 1] = z[:n]
    return out
