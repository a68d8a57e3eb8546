"""Batched API over libbrov2.so: RHS, rollouts, sliding-window endpoint RMSE, EDMDc lift / Gram.

Host path  : NumPy arrays in / out (the library stages them through HBM).
Device path: device-resident arrays -- either DevArray (this module: brov_malloc'ed HBM, no torch anywhere; what the drop-in
             classes use) or torch CUDA tensors (fp64, contiguous; the caller's own pipeline, bench.py, dist.py).  Only the device
             address and the stream cross the C ABI; results stay in HBM.  The two kinds run the same launches with the same
             arguments: results are bit-identical.
"""
import ctypes
import os

import numpy as np

from . import _lib
try:                                    # optional CPython helper (csrc/bagtable.c, built by _build.build_bagtable): header walk of array lists
    from . import _bagtable
except ImportError:
    _bagtable = None
from ._lib import (EULER, RK4, LAG_PER_CALL, LAG_PER_STEP, LAYOUT_BTU, LAYOUT_TUB, LAYOUT_TPB, THRUSTER_EULER, WRENCH_EULER,
                   WRENCH_QUAT, DIST_IID_UNIFORM, DIST_AR1, NX, NU, as_f64, _hptr, default_context)

__all__ = ["rhs", "thruster_forces", "rollout", "window_endpoint_se", "window_rmse", "rollout_dev", "fill_controls_dev",
           "window_endpoint_se_dev", "lift", "gram", "gram_dev", "gram_ragged_dev", "pinv_apply_ragged_dev", "upload_bags", "BagTable", "solve_AB", "solve_AB_fit_order", "pinv_apply", "pinv_apply_dev", "fit_dev", "apply_decomposition", "gtg_decomposition", "kmeans_lloyd", "kmeans_centers", "kmeans_centers_dev", "multistep_se", "multistep_se_linear", "simulate_lifted", "DevArray", "col_stats_dev"]

INTEGRATORS = {"euler": EULER, "rk4": RK4, EULER: EULER, RK4: RK4}
LAYOUTS = {"btu": LAYOUT_BTU, "tub": LAYOUT_TUB, "tpb": LAYOUT_TPB, LAYOUT_BTU: LAYOUT_BTU, LAYOUT_TUB: LAYOUT_TUB, LAYOUT_TPB: LAYOUT_TPB}


def _dims(U_shape, lay):
    """(B, T, channels) of a control / trajectory array of the given layout ([B,T,c], [T,c,B] or [T,ceil(c/2),B,2])."""
    if lay == LAYOUT_BTU:
        return U_shape[0], U_shape[1], U_shape[2]
    if lay == LAYOUT_TUB:
        return U_shape[2], U_shape[0], U_shape[1]
    return U_shape[2], U_shape[0], 2 * U_shape[1]


def _shape(lay, B, rows, c):
    if lay == LAYOUT_BTU:
        return (B, rows, c)
    if lay == LAYOUT_TUB:
        return (rows, c, B)
    return (rows, (c + 1) // 2, B, 2)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DevArray:
    """A row-major array in HBM owned through the C ABI (brov_malloc / brov_free, copies on the ctx's stream): the device-resident
    operand of the `_dev` entry points for callers without torch.  float64 unless said otherwise; `view` / `reshape` give
    non-owning aliases that keep their base alive."""
    __slots__ = ("ctx", "ptr", "shape", "dtype", "_base", "_owned")
    is_cuda = True

    def __init__(self, ctx, shape, dtype=np.float64, _ptr=None, _base=None):
        self.ctx = ctx
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self._base = _base
        self._owned = _ptr is None
        if _ptr is None:
            p_ = ctypes.c_void_p()
            ctx.check(ctx.lib.brov_malloc(ctx.h, max(self.nbytes, 8), ctypes.byref(p_)), "brov_malloc")
            _ptr = p_.value
        self.ptr = _ptr

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def data_ptr(self):
        return self.ptr

    def dim(self):
        return len(self.shape)

    def stride(self, i):
        return int(np.prod(self.shape[i + 1:], dtype=np.int64))

    def is_contiguous(self):
        return True

    @classmethod
    def from_host(cls, ctx, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        return cls(ctx, a.shape, dtype).copy_from_host(a)

    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, "size mismatch"
        if a.nbytes:
            self.ctx.check(self.ctx.lib.brov_memcpy_h2d(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes), "brov_memcpy_h2d")
        return self

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            self.ctx.check(self.ctx.lib.brov_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes), "brov_memcpy_d2h")
        return out

    def zero_(self):
        self.ctx.check(self.ctx.lib.brov_memset(self.ctx.h, self.ptr, 0, self.nbytes), "brov_memset")
        return self

    def view(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        shape = list(int(v) for v in shape)
        total = self.numel()
        if -1 in shape:
            i = shape.index(-1)
            rest = int(np.prod([v for v in shape if v != -1], dtype=np.int64))
            shape[i] = total // rest if rest else 0
        assert int(np.prod(shape, dtype=np.int64)) == total, f"cannot view {self.shape} as {tuple(shape)}"
        return DevArray(self.ctx, tuple(shape), self.dtype, _ptr=self.ptr, _base=self)

    reshape = view

    def rows(self, a, b):
        """alias of rows a .. b-1 (leading dimension)"""
        a, b = int(a), int(b)
        assert 0 <= a <= b <= self.shape[0]
        return DevArray(self.ctx, (b - a,) + self.shape[1:], self.dtype, _ptr=self.ptr + a * self.stride(0) * self.dtype.itemsize, _base=self)

    def free(self):
        if self._owned and self.ptr:
            p_, self.ptr = self.ptr, None
            self.ctx.lib.brov_free(self.ctx.h, ctypes.c_void_p(p_))

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class _NativeArrays:
    """allocation / movement of DevArray operands (no torch); launches go to the ctx's own stream (the null stream)"""
    kind = "native"

    def __init__(self, ctx):
        self.ctx = ctx

    def bind(self):
        self.ctx.use_null_stream()

    def empty(self, shape, dtype=np.float64):
        return DevArray(self.ctx, shape, dtype)

    def upload(self, a, dtype=np.float64):
        return DevArray.from_host(self.ctx, a, dtype)

    def download(self, d):
        return d.numpy()

    def sync(self):
        self.ctx.sync()

    def mem_free(self):
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.ctx.check(self.ctx.lib.brov_mem_info(self.ctx.h, ctypes.byref(f), ctypes.byref(t)), "brov_mem_info")
        return int(f.value)


class _TorchArrays:
    """the same for torch CUDA tensors; launches go to torch's current stream"""
    kind = "torch"

    def __init__(self, ctx):
        import torch
        self.ctx, self.torch = ctx, torch
        self.dev = torch.device("cuda", ctx.device)

    def bind(self):
        self.ctx.use_torch_stream()

    def empty(self, shape, dtype=np.float64):
        t = self.torch
        return t.empty(shape, dtype={np.dtype(np.float64): t.float64, np.dtype(np.int32): t.int32, np.dtype(np.uint8): t.uint8}[np.dtype(dtype)], device=self.dev)

    def upload(self, a, dtype=np.float64):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.dev)

    def download(self, d):
        return d.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def mem_free(self):
        return int(self.torch.cuda.mem_get_info(self.dev)[0])


def arrays_of(x, ctx):
    """the array namespace an operand belongs to"""
    return _TorchArrays(ctx) if _is_torch(x) else _NativeArrays(ctx)


def _ctx_of(x, ctx):
    if ctx is not None:
        return ctx
    return x.ctx if isinstance(x, DevArray) else default_context(x.device.index)


def _bind(x, ctx):
    """launch where the operand lives: DevArray -> the ctx's null stream, torch tensor -> torch's current stream"""
    if isinstance(x, DevArray):
        ctx.use_null_stream()
    else:
        ctx.use_torch_stream()


def _dptr(t):
    """device pointer of a DevArray or of a contiguous fp64 CUDA tensor (or None)."""
    if t is None:
        return None
    if isinstance(t, DevArray):
        assert t.dtype == np.float64, "need a float64 DevArray"
        return t.ptr
    import torch
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need a contiguous fp64 CUDA tensor"
    return t.data_ptr()


def _drows(t):
    """device pointer of a 2-D fp64 array whose ROWS are contiguous (any row stride: a column slice of a padded buffer)."""
    if isinstance(t, DevArray):
        assert t.dtype == np.float64 and t.ndim == 2
        return t.ptr
    import torch
    assert t.is_cuda and t.dtype == torch.float64 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1], \
        "need a 2-D fp64 CUDA tensor with contiguous rows"
    return t.data_ptr()


# ------------------------------------------------------------------------------------------ host path
def rhs(model, x, u, dt=0.02, lag=None, ctx=None):
    """Batched dynamics().  Returns (xdot [B,nx], lag_after [B,8,3] or None)."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    x = as_f64(x).reshape(-1, NX[model])
    u = as_f64(u).reshape(-1, NU[model])
    B = x.shape[0]
    assert u.shape[0] == B
    out = np.empty_like(x)
    lag_io = None
    if model == THRUSTER_EULER:
        lag_io = np.zeros((B, 8, 3)) if lag is None else as_f64(lag).reshape(B, 8, 3).copy()
    ctx.check(ctx.lib.brov_rhs(ctx.h, model, B, _hptr(x), _hptr(u), float(dt), _hptr(lag_io), _hptr(out)), "brov_rhs")
    return out, lag_io


def thruster_forces(u, dt=0.02, lag=None, ctx=None):
    """Batched compute_thruster_forces().  Returns (tau [B,6], lag_after [B,8,3])."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    u = as_f64(u).reshape(-1, 8)
    B = u.shape[0]
    lag_io = np.zeros((B, 8, 3)) if lag is None else as_f64(lag).reshape(B, 8, 3).copy()
    tau = np.empty((B, 6))
    ctx.check(ctx.lib.brov_thruster_forces(ctx.h, B, _hptr(u), float(dt), _hptr(lag_io), _hptr(tau)), "brov_thruster_forces")
    return tau, lag_io


def rollout(model, integrator, x0, U, dt, lag=None, lag_mode=LAG_PER_CALL, layout="btu", stride=1, store=True, ctx=None,
            return_lag=True):
    """simulate_physics over a batch (host arrays).

    x0 [B,nx]; U [B,T,nu] (layout "btu") or [T,nu,B] ("tub").
    Returns dict(traj, xT [B,nx], lag [B,8,3] | None); traj is [B,T//stride+1,nx] or [T//stride+1,nx,B].
    return_lag=False with lag=None starts from a zero lag state and skips the per-thruster bookkeeping."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    integ, lay = INTEGRATORS[integrator], LAYOUTS[layout]
    nx, nu = NX[model], NU[model]
    U = as_f64(U)
    B, T, nu_ = _dims(U.shape, lay)
    assert nu_ == (nu + 1) // 2 * 2 if lay == LAYOUT_TPB else nu_ == nu, f"U has {nu_} channels, model needs {nu}"
    x0 = as_f64(x0).reshape(B, nx)
    lag_io = None
    if model == THRUSTER_EULER and (lag is not None or return_lag):
        lag_io = np.zeros((B, 8, 3)) if lag is None else as_f64(lag).reshape(B, 8, 3).copy()
    rows = T // stride + 1
    traj = None
    if store:
        traj = np.empty(_shape(lay, B, rows, nx))
    xT = np.empty((B, nx))
    ctx.check(ctx.lib.brov_rollout(ctx.h, model, integ, lag_mode, lay, B, T, float(dt), _hptr(x0), _hptr(U), _hptr(lag_io),
                                   _hptr(traj), int(stride), _hptr(xT)), "brov_rollout")
    return dict(traj=traj, xT=xT, lag=lag_io)


def window_endpoint_se(model, integrator, X, U, H, dt, carry_lag=True, ctx=None):
    """Sum of squared endpoint errors over all sliding windows; returns (se_total, per_window[N-H])."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    X = as_f64(X).reshape(-1, NX[model])
    U = as_f64(U).reshape(-1, NU[model])
    N = X.shape[0]
    assert U.shape[0] >= N, "U must be aligned with X"
    per = np.zeros(max(N - H, 0))
    se = ctypes.c_double(0.0)
    ctx.check(ctx.lib.brov_window_endpoint_se(ctx.h, model, INTEGRATORS[integrator], N, int(H), float(dt), _hptr(X), _hptr(U),
                                              int(bool(carry_lag)), ctypes.addressof(se), _hptr(per) if len(per) else None),
              "brov_window_endpoint_se")
    return se.value, per


def window_rmse(model, integrator, X, U, H, dt, carry_lag=True, ctx=None):
    """multistep_rmse_endpoint_physics (training/train_tank_brov2_full_comparison.py:469-487)."""
    X = np.asarray(X)
    n_start = len(X) - H
    if n_start <= 0:
        return float("nan")
    se, _ = window_endpoint_se(model, integrator, X, U, H, dt, carry_lag, ctx)
    return float(np.sqrt(se / (n_start * NX[model])))


# ------------------------------------------------------------------------------------------ device path
def rollout_dev(model, integrator, x0, U, dt, lag=None, traj=None, xT=None, lag_mode=LAG_PER_CALL, layout="tub", stride=1,
                ctx=None):
    """Asynchronous rollout on device-resident arrays (DevArray or torch CUDA tensors; no copies).  Shapes as in rollout();
    traj / xT / lag are written in place when given."""
    ctx = _ctx_of(x0, ctx)
    _bind(x0, ctx)
    lay = LAYOUTS[layout]
    nu = NU[model]
    B, T, nu_ = _dims(tuple(U.shape), lay)
    assert (nu_ == (nu + 1) // 2 * 2 if lay == LAYOUT_TPB else nu_ == nu) and tuple(x0.shape) == (B, NX[model])
    if traj is not None:
        rows = T // stride + 1
        want = _shape(lay, B, rows, NX[model])
        assert tuple(traj.shape) == want, f"traj shape {tuple(traj.shape)} != {want}"
    ctx.check(ctx.lib.brov_rollout_dev(ctx.h, model, INTEGRATORS[integrator], lag_mode, lay, B, T, float(dt), _dptr(x0), _dptr(U),
                                       _dptr(lag), _dptr(traj), int(stride), _dptr(xT)), "brov_rollout_dev")


def fill_controls_dev(U, layout, dist="iid", seed=0x5EED, b0=0, T_total=None, scale=None, ctx=None):
    """Fill a device-resident U ([B,T,nu] or [T,nu,B]; DevArray or torch CUDA tensor) with the synthetic control stream."""
    ctx = _ctx_of(U, ctx)
    _bind(U, ctx)
    lay = LAYOUTS[layout]
    B, T, nu = _dims(tuple(U.shape), lay)
    d = {"iid": DIST_IID_UNIFORM, "ar1": DIST_AR1}[dist]
    sc = None if scale is None else as_f64(scale).reshape(nu)
    ctx.check(ctx.lib.brov_fill_controls_dev(ctx.h, lay, d, B, T, nu, ctypes.c_uint64(seed), int(b0), int(T_total or T),
                                             _hptr(sc), _dptr(U)), "brov_fill_controls_dev")


def window_endpoint_se_dev(model, integrator, X, U, H, dt, se_total, per_window, carry_lag=True, ctx=None):
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    N = X.shape[0]
    ctx.check(ctx.lib.brov_window_endpoint_se_dev(ctx.h, model, INTEGRATORS[integrator], N, int(H), float(dt), _dptr(X), _dptr(U),
                                                  int(bool(carry_lag)), _dptr(se_total), _dptr(per_window)),
              "brov_window_endpoint_se_dev")


# ------------------------------------------------------------------------------------------ EDMDc
def lift(X, C, gamma, ctx=None):
    """phi(X) = [X, rbf(X)]  (KoopmanEDMDc._lift); X [N,n] -> [N,n+k]."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    X = as_f64(X)
    C = as_f64(C)
    N, n = X.shape
    k = C.shape[0]
    Z = np.empty((N, n + k))
    ctx.check(ctx.lib.edmdc_lift(ctx.h, N, n, k, float(gamma), _hptr(X), _hptr(C), _hptr(Z)), "edmdc_lift")
    return Z


class BagTable:
    """The bookkeeping of a trajectory list (fit_multi's X_list / U_list, Koopman/koopmanEDMDc.py:113-152) for brov_upload_bags and the
    ragged Gram: `offsets` int64 [nbags + 1] of the stacked states, `lens` = rows per bag, `u_rows` = rows of each U that go next to them
    (U is stored ROW-ALIGNED with X; like the reference -- `U[:-1]` next to `X[:-1]` -- a bag needs len(U) >= len(X) - 1, rows past
    len(X) are ignored), and the host addresses of the bags (C-contiguous fp64; anything else is converted and kept alive here)."""

    def __init__(self, X_list, U_list, n, r):
        nb = len(X_list)
        self.keep = []
        if _bagtable is not None and nb > 0:
            self._init_native(X_list, U_list, n, r, nb)
            return
        nd, f64 = np.ndarray, np.float64
        lens, urows, px, pu = [0] * nb, [0] * nb, [0] * nb, [0] * nb
        for b in range(nb):                      # ~1.3 us per bag: one __array_interface__ dict per array carries dtype, layout, shape, address
            X, U = X_list[b], U_list[b]
            ai = X.__array_interface__ if type(X) is nd else None
            if ai is None or ai["typestr"] != "<f8" or ai["strides"] is not None:
                X = np.ascontiguousarray(X, dtype=f64)
                self.keep.append(X)
                ai = X.__array_interface__
            au = U.__array_interface__ if type(U) is nd else None
            if au is None or au["typestr"] != "<f8" or au["strides"] is not None:
                U = np.ascontiguousarray(U, dtype=f64)
                self.keep.append(U)
                au = U.__array_interface__
            sx, su = ai["shape"], au["shape"]
            if len(sx) != 2 or sx[1] != n or len(su) != 2 or su[1] != r:
                raise AssertionError(f"bag {b}: X {sx} / U {su} do not match state_dim {n} / input_dim {r}")
            lx, lu = sx[0], su[0]
            if lx >= 2 and lu < lx - 1:
                raise ValueError(f"bag {b}: U has {lu} rows, needs at least len(X) - 1 = {lx - 1}")
            lens[b], urows[b] = lx, (lu if lu < lx else lx)
            px[b], pu[b] = ai["data"][0], au["data"][0]
        self.lens = np.array(lens, dtype=np.int64)
        self.u_rows = np.array(urows, dtype=np.int64)
        self.x_ptrs = np.array(px, dtype=np.uint64)
        self.u_ptrs = np.array(pu, dtype=np.uint64)
        self._finish(n, r, nb)

    def _init_native(self, X_list, U_list, n, r, nb):
        """The same table through the buffer protocol (csrc/bagtable.c): ~60 ns per array instead of ~0.7 us."""
        f64 = np.float64
        out = []
        for seq, ncols in ((X_list, n), (U_list, r)):
            ptrs, rows = np.zeros(nb, dtype=np.uint64), np.zeros(nb, dtype=np.int64)
            seq = seq if isinstance(seq, (list, tuple)) else list(seq)
            start = 0
            while True:
                bad = _bagtable.fill(seq, ncols, start, ptrs, rows)
                if bad < 0:
                    break
                a_ = np.ascontiguousarray(seq[bad], dtype=f64)           # not a C-contiguous float64 array (or the wrong shape): convert, check
                if a_.ndim != 2 or a_.shape[1] != ncols:
                    raise AssertionError(f"bag {bad}: array of shape {a_.shape} does not match state_dim {n} / input_dim {r}")
                self.keep.append(a_)
                if seq is X_list or seq is U_list:
                    seq = list(seq)
                seq[bad] = a_
                start = bad
            self.keep.append(seq)                                         # the (possibly patched) list keeps its arrays alive
            out.append((ptrs, rows))
        (self.x_ptrs, self.lens), (self.u_ptrs, urows) = out
        short = (self.lens >= 2) & (urows < self.lens - 1)
        if short.any():
            b = int(np.argmax(short))
            raise ValueError(f"bag {b}: U has {int(urows[b])} rows, needs at least len(X) - 1 = {int(self.lens[b]) - 1}")
        self.u_rows = np.minimum(urows, self.lens)
        self._finish(n, r, nb)

    def _finish(self, n, r, nb):
        self.n, self.r, self.nbags = n, r, nb
        self.offsets = np.zeros(nb + 1, dtype=np.int64)
        np.cumsum(self.lens, out=self.offsets[1:])
        self.rows = int(self.offsets[-1])
        self.pairs = int(np.maximum(self.lens - 1, 0).sum())

    def upload_into(self, dX, dU, ctx):
        """brov_upload_bags twice: the states of all bags into dX [rows, n], the inputs row-aligned into dU [rows, r] (device pointers)."""
        if self.nbags == 0 or self.rows == 0:
            return
        dst = np.ascontiguousarray(self.offsets[:-1])
        ctx.check(ctx.lib.brov_upload_bags(ctx.h, self.nbags, self.x_ptrs.ctypes.data, self.lens.ctypes.data, dst.ctypes.data, self.n, dX),
                  "brov_upload_bags")
        if self.r:
            ctx.check(ctx.lib.brov_upload_bags(ctx.h, self.nbags, self.u_ptrs.ctypes.data, self.u_rows.ctypes.data, dst.ctypes.data, self.r, dU),
                      "brov_upload_bags")


def upload_bags(X_list, U_list, n, r, device=None, ctx=None, arrays="native"):
    """fit_multi's trajectory list in HBM, uploaded once: (Xd [rows, n], Ud [rows, r] row-aligned with Xd, offsets int64 [nbags + 1]).
    Xd is np.vstack(X_list) (Koopman/koopmanEDMDc.py:125); no stacked copy is formed on the host.  arrays: "native" = DevArray (no torch),
    "torch" = torch CUDA tensors."""
    ctx = ctx or default_context(device)
    ns = _TorchArrays(ctx) if arrays == "torch" else _NativeArrays(ctx)
    ns.bind()
    bt = BagTable(X_list, U_list, n, r)
    rows = bt.rows
    Xd = ns.empty((rows, n))
    # rows of U that no bag provides (len(U) == len(X) - 1) stay as allocated: the kernels never read the input next to a bag's last state
    Ud = ns.empty((rows, r))
    bt.upload_into(Xd.data_ptr(), Ud.data_ptr(), ctx)
    return Xd, Ud, bt.offsets


class _DevBuf:
    """brov_malloc'ed scratch of the torch-free host entry points (freed on exit)."""

    def __init__(self, ctx, *sizes):
        self.ctx, self.ptrs = ctx, []
        try:
            for nbytes in sizes:
                p_ = ctypes.c_void_p()
                ctx.check(ctx.lib.brov_malloc(ctx.h, max(int(nbytes), 8), ctypes.byref(p_)), "brov_malloc")
                self.ptrs.append(p_.value)
        except Exception:
            self.close()
            raise

    def close(self):
        for p_ in self.ptrs:
            self.ctx.lib.brov_free(self.ctx.h, p_)
        self.ptrs = []

    def __enter__(self):
        return self.ptrs

    def __exit__(self, *exc):
        self.close()
        return False


def gram(X_list, U_list, C, gamma, ctx=None):
    """G^T G [p,p] and G^T Y [p,d] over bags (no cross-bag pairs) -- fit / fit_multi normal equations
    (Koopman/koopmanEDMDc.py:89-97,129-147).  Host arrays in and out; the whole list is uploaded once (brov_upload_bags) and goes
    through ONE ragged Gram call (edmdc_gram_ragged_dev), whatever the number of bags."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    C = as_f64(C)
    k, n = C.shape
    r = np.asarray(U_list[0]).shape[1]
    d, p = n + k, n + k + r
    bt = BagTable(X_list, U_list, n, r)
    rows, off, npairs = bt.rows, bt.offsets, bt.pairs
    GtG = np.zeros((p, p))
    GtY = np.zeros((p, d))
    with _DevBuf(ctx, rows * n * 8, rows * r * 8, k * n * 8, p * p * 8, p * d * 8) as (dX, dU, dC, dG, dY):
        bt.upload_into(dX, dU, ctx)
        ctx.check(ctx.lib.brov_memcpy_h2d(ctx.h, dC, _hptr(C), k * n * 8), "brov_memcpy_h2d")
        ctx.check(ctx.lib.edmdc_gram_ragged_dev(ctx.h, n, r, k, float(gamma), dC, bt.nbags, off.ctypes.data, dX, dU, 0, dG, dY),
                  "edmdc_gram_ragged_dev")
        ctx.check(ctx.lib.brov_memcpy_d2h(ctx.h, _hptr(GtG), dG, p * p * 8), "brov_memcpy_d2h")
        ctx.check(ctx.lib.brov_memcpy_d2h(ctx.h, _hptr(GtY), dY, p * d * 8), "brov_memcpy_d2h")
    return GtG, GtY, npairs


def gram_decomposition(n, r, k):
    """(ntasks, nslabs) of the device Gram for a shape: ntasks blocks of 4 x 6 tiles of 16 x 16 outputs per slab of rows."""
    nt, ns = ctypes.c_int(0), ctypes.c_int(0)
    rc = _lib.load_library().edmdc_gram_decomposition(int(n), int(r), int(k), ctypes.byref(nt), ctypes.byref(ns))
    if rc:
        raise ValueError(f"edmdc_gram_decomposition: unsupported shape n={n} r={r} k={k}")
    return nt.value, ns.value


def apply_decomposition(n, r, k):
    """Work decomposition of edmdc_pinv_apply for a shape: dict(wrows_items_per_192_rows, wrows_tiles_wanted, wty_tasks,
    wty_slabs) -- see include/brov2.h (edmdc_apply_decomposition)."""
    v = [ctypes.c_int(0) for _ in range(4)]
    rc = _lib.load_library().edmdc_apply_decomposition(int(n), int(r), int(k), *[ctypes.byref(x) for x in v])
    if rc:
        raise ValueError(f"edmdc_apply_decomposition: unsupported shape n={n} r={r} k={k}")
    return dict(zip(("wrows_items_per_192_rows", "wrows_tiles_wanted", "wty_tasks", "wty_slabs"), (x.value for x in v)))


def gtg_decomposition(n, r, k):
    """(ntasks, nslabs) of the G^T G-only Gram (gram_dev with GtY=None: fit()'s Gram pass)."""
    nt, ns = ctypes.c_int(0), ctypes.c_int(0)
    rc = _lib.load_library().edmdc_gtg_decomposition(int(n), int(r), int(k), ctypes.byref(nt), ctypes.byref(ns))
    if rc:
        raise ValueError(f"edmdc_gtg_decomposition: unsupported shape n={n} r={r} k={k}")
    return nt.value, ns.value


def gram_dev(X, U, C, gamma, nbags, L, x_bag_stride, u_bag_stride, GtG, GtY, accumulate=False, ctx=None):
    """Device Gram on device-resident arrays: X [rows,n] states, U [rows,r] inputs in bag layout (see include/brov2.h).
    GtY=None: G^T G alone -- all KoopmanEDMDc.fit needs before its pinv (a third of the tile products)."""
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    n, k, r = X.shape[-1], C.shape[0], U.shape[-1]
    ctx.check(ctx.lib.edmdc_gram_dev(ctx.h, n, r, k, float(gamma), _dptr(C), int(nbags), int(L), int(x_bag_stride), int(u_bag_stride),
                                     _dptr(X), _dptr(U), int(bool(accumulate)), _dptr(GtG), _dptr(GtY)), "edmdc_gram_dev")


def _offsets(bag_offsets):
    off = np.ascontiguousarray(np.asarray(bag_offsets, dtype=np.int64))
    assert off.ndim == 1 and off.size >= 1
    return off


def gram_ragged_dev(X, U, C, gamma, bag_offsets, GtG, GtY, accumulate=False, ctx=None):
    """Device Gram over a RAGGED bag list (fit_multi): X [rows,n] the stacked states, U [rows,r] row-aligned with X, bag b = rows
    bag_offsets[b] .. bag_offsets[b+1]-1 (host int64 array) -- see edmdc_gram_ragged_dev in include/brov2.h.  GtY=None: G^T G alone."""
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    off = _offsets(bag_offsets)
    n, k, r = X.shape[-1], C.shape[0], U.shape[-1]
    assert X.shape[0] == off[-1] and U.shape[0] == off[-1], "X / U rows must equal bag_offsets[-1]"
    ctx.check(ctx.lib.edmdc_gram_ragged_dev(ctx.h, n, r, k, float(gamma), _dptr(C), off.size - 1, off.ctypes.data, _dptr(X), _dptr(U),
                                            int(bool(accumulate)), _dptr(GtG), _dptr(GtY)), "edmdc_gram_ragged_dev")


def pinv_apply_ragged_dev(X, U, C, gamma, bag_offsets, P, M, ctx=None):
    """pinv_apply_dev for the ragged bag list of gram_ragged_dev."""
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    off = _offsets(bag_offsets)
    n, k, r = X.shape[-1], C.shape[0], U.shape[-1]
    P = as_f64(P)
    assert P.shape == (n + k + r, n + k + r) and tuple(M.shape) == (n + k + r, n + k)
    ctx.check(ctx.lib.edmdc_pinv_apply_ragged_dev(ctx.h, n, r, k, float(gamma), _dptr(C), off.size - 1, off.ctypes.data, _dptr(X), _dptr(U),
                                                  _hptr(P), _dptr(M)), "edmdc_pinv_apply_ragged_dev")


def kmeans_lloyd(X, C_init, max_iter=300, tol_abs=0.0, mean=None, ctx=None):
    """Lloyd iterations on the GPU (edmdc_kmeans_lloyd): returns (centres [k,n] in the frame of X - mean,
    labels [N] int32, inertia, n_iter)."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    X = as_f64(X)
    C = as_f64(C_init).copy()
    N, n = X.shape
    k = C.shape[0]
    m = None if mean is None else as_f64(mean).reshape(n)
    labels = np.empty(N, dtype=np.int32)
    inertia = ctypes.c_double(0.0)
    n_iter = ctypes.c_int(0)
    ctx.check(ctx.lib.edmdc_kmeans_lloyd(ctx.h, N, n, k, _hptr(X), _hptr(m), _hptr(C), int(max_iter), float(tol_abs),
                                         labels.ctypes.data, ctypes.byref(inertia), ctypes.byref(n_iter)), "edmdc_kmeans_lloyd")
    return C, labels, inertia.value, n_iter.value


KMEANSPP_MAX_ROWS = 30_000_000      # edmdc_kmeanspp_dev's limit (include/brov2.h)


def kmeanspp_draws(N, k, random_state=0):
    """The random numbers scikit-learn's `_kmeans_plusplus` consumes, in its order, from numpy's legacy RandomState:
    (first_index, uniforms [(k-1), n_trials], n_trials).  `choice(N, p=uniform)` is evaluated the way numpy does
    (cumsum of p, normalise, searchsorted side='right' of one random_sample)."""
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    L = 2 + int(np.log(k))
    # choice(N, p=1/N): cdf = cumsum(p) / cdf[-1], index = searchsorted(cdf, one random_sample, side="right") = floor(u N)
    # unless u N lies within the rounding error of the sequential cumsum (<= 2 N^2 eps index units) of an integer: only then
    # is the N-element cdf actually formed (0.05-2 s of host time at N = 1e7, as long as the whole device seeding)
    u0 = rs.random_sample()
    t = u0 * N
    thr = 4.0 * float(N) * float(N) * 2.3e-16
    if thr < 0.25 and thr < t - np.floor(t) < 1.0 - thr:
        first = int(t)
    else:
        cdf = np.full(N, 1.0 / N).cumsum()
        cdf /= cdf[-1]
        first = int(cdf.searchsorted(u0, side="right"))
    U = np.empty((max(k - 1, 0), L))
    for c in range(k - 1):
        U[c] = rs.uniform(size=L)
    return first, U, L


def kmeanspp_dev(X, k, mean=None, random_state=0, ctx=None, n_global=None):
    """k-means++ seeding of a device-resident X ([N,n] DevArray or torch CUDA tensor) with scikit-learn's algorithm and random stream
    (edmdc_kmeanspp_dev).  Returns (C0 [k,n] device array of X's kind, in the frame of X - mean; indices [k] int64 numpy).
    n_global: rows over all ranks when X is this rank's shard of a sharded seeding (dist.kmeanspp_sharded installs the exchange):
    the random numbers are drawn for that many rows, indices come back global."""
    ctx = _ctx_of(X, ctx)
    ns = arrays_of(X, ctx)
    ns.bind()
    N, n = X.shape
    assert X.stride(1) == 1
    first, U, L = kmeanspp_draws(N if n_global is None else int(n_global), k, random_state)
    C = ns.empty((k, n))
    ind = np.empty(k, dtype=np.int64)
    m = None if mean is None else as_f64(mean).reshape(n)
    Uc = np.ascontiguousarray(U)
    ns.sync()
    ctx.check(ctx.lib.edmdc_kmeanspp_dev(ctx.h, N, n, k, _drows(X), X.stride(0), _hptr(m), first, L, _hptr(Uc) if k > 1 else None,
                                         _dptr(C), ind.ctypes.data), "edmdc_kmeanspp_dev")
    return C, ind


def col_stats_dev(X, ctx=None):
    """(mean [n], var [n]) of the columns of a device-resident X [N,n] as host arrays (edmdc_col_stats_dev): scikit-learn's
    `X.mean(axis=0)` / `np.var(X, axis=0)` before its Lloyd loop.  The same kernel and summation order for DevArray and torch operands."""
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    N, n = X.shape
    mean, var = np.empty(n), np.empty(n)
    ctx.check(ctx.lib.edmdc_col_stats_dev(ctx.h, N, n, _drows(X), X.stride(0), mean.ctypes.data, var.ctypes.data), "edmdc_col_stats_dev")
    return mean, var


def kmeans_centers(X, k, random_state=0, max_iter=300, tol=1e-4, init="hip", ctx=None):
    """RBF centres the way KoopmanEDMDc.fit gets them (sklearn KMeans(k, n_init="auto", random_state=0),
    Koopman/koopmanEDMDc.py:85): k-means++ seeding with scikit-learn's random stream, then Lloyd's E/M loop with
    scikit-learn's stopping rules, both on the GPU over the full data.  init="sklearn" seeds with
    sklearn.cluster.kmeans_plusplus on the host instead (same centres; kept for cross-checks).  X: host array [N,n]."""
    X = as_f64(X)
    ctx = ctx or default_context()
    Xd = DevArray.from_host(ctx, X)
    C, _, _ = kmeans_centers_dev(Xd, k, random_state=random_state, max_iter=max_iter, tol=tol, init=init, ctx=ctx)
    return C.numpy()


def kmeans_centers_dev(X, k, random_state=0, max_iter=300, tol=1e-4, init="hip", init_rows=None, ctx=None, timings=None):
    """kmeans_centers for a device-resident X ([N,n] DevArray or torch CUDA tensor).  Returns (centres [k,n] device array of X's
    kind, inertia, n_iter).  init_rows: seed on a seeded subsample of that many rows instead of all N (not what scikit-learn does;
    useful with init="sklearn", whose host seeding takes minutes at N = 1e7, and applied automatically -- 1.5e7 rows --
    beyond the 3e7 rows the device seeding accepts); the Lloyd iterations always run over all N rows.  timings: dict that receives
    kmeanspp_ms / lloyd_ms (kernel time, needs ctx.set_timing(True)) and host_draws_s."""
    ctx = _ctx_of(X, ctx)
    ns = arrays_of(X, ctx)
    ns.bind()
    N, n = X.shape
    mean_h, var_h = col_stats_dev(X, ctx=ctx)
    tol_abs = float(var_h.mean() * tol)
    Xi = X
    if init_rows is None and N > KMEANSPP_MAX_ROWS:
        init_rows = KMEANSPP_MAX_ROWS // 2          # the device seeding holds its running-sum table in LDS: N <= 3e7 rows
        import warnings
        warnings.warn(f"kmeans_centers_dev: {N} rows exceed the {KMEANSPP_MAX_ROWS} the device k-means++ seeding accepts; seeding on a seeded "
                      f"subsample of {init_rows} rows (scikit-learn would seed on all rows: the centres differ from its); the Lloyd "
                      "iterations run over all rows", RuntimeWarning, stacklevel=2)
    if init_rows is not None and N > init_rows:
        idx = np.random.RandomState(random_state).choice(N, init_rows, replace=False)
        if ns.kind == "torch":
            Xi = X[ns.torch.from_numpy(idx).to(X.device)].contiguous()
        else:
            Xi = ns.upload(ns.download(X)[idx])          # (a host round trip: the subsample path is a fallback beyond 3e7 rows)
    if init == "hip":
        C, _ = kmeanspp_dev(Xi, k, mean=mean_h, random_state=random_state, ctx=ctx)
        if timings is not None and getattr(ctx, "timing", False):
            timings["kmeanspp_ms"] = ctx.last_kernel_ms()
    elif init == "sklearn":
        from sklearn.cluster import kmeans_plusplus
        C0, _ = kmeans_plusplus(ns.download(Xi) - mean_h, k, random_state=np.random.RandomState(random_state))
        C = ns.upload(C0)
    else:
        raise ValueError("init must be 'hip' or 'sklearn'")
    labels = ns.empty((N,), np.int32)
    inertia = ctypes.c_double(0.0)
    n_iter = ctypes.c_int(0)
    ns.sync()
    ctx.check(ctx.lib.edmdc_kmeans_lloyd_dev(ctx.h, N, n, k, _drows(X), X.stride(0), _hptr(mean_h), _dptr(C), int(max_iter), tol_abs,
                                             labels.data_ptr(), ctypes.byref(inertia), ctypes.byref(n_iter)), "edmdc_kmeans_lloyd_dev")
    if timings is not None and getattr(ctx, "timing", False):
        timings["lloyd_ms"] = ctx.last_kernel_ms()
    # back to the caller's frame: k x n values through the host (IEEE addition either way: the same bits as a device add)
    Cf = ns.upload(ns.download(C) + mean_h)
    return Cf, inertia.value, n_iter.value


def pinv_apply(X_list, U_list, C, gamma, P, ctx=None):
    """M [p,d] = (P G^T) Y evaluated in that order over bags (KoopmanEDMDc.fit's association,
    Koopman/koopmanEDMDc.py:97), P [p,p] = pinv(G^T G + ridge I) from the host.  One upload, one ragged call."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    C = as_f64(C)
    P = as_f64(P)
    k, n = C.shape
    r = np.asarray(U_list[0]).shape[1]
    d, p = n + k, n + k + r
    assert P.shape == (p, p)
    bt = BagTable(X_list, U_list, n, r)
    rows, off = bt.rows, bt.offsets
    M = np.zeros((p, d))
    with _DevBuf(ctx, rows * n * 8, rows * r * 8, k * n * 8, p * d * 8) as (dX, dU, dC, dM):
        bt.upload_into(dX, dU, ctx)
        ctx.check(ctx.lib.brov_memcpy_h2d(ctx.h, dC, _hptr(C), k * n * 8), "brov_memcpy_h2d")
        ctx.check(ctx.lib.edmdc_pinv_apply_ragged_dev(ctx.h, n, r, k, float(gamma), dC, bt.nbags, off.ctypes.data, dX, dU, _hptr(P), dM),
                  "edmdc_pinv_apply_ragged_dev")
        ctx.check(ctx.lib.brov_memcpy_d2h(ctx.h, _hptr(M), dM, p * d * 8), "brov_memcpy_d2h")
    return M


def pinv_apply_dev(X, U, C, gamma, nbags, L, x_bag_stride, u_bag_stride, P, M, ctx=None):
    """Device form of pinv_apply on device-resident arrays (bag layout as gram_dev): P [p,p] host array (the host's pinv), M [p,d]
    device array, overwritten with (P G^T) Y."""
    ctx = _ctx_of(X, ctx)
    _bind(X, ctx)
    n, k, r = X.shape[-1], C.shape[0], U.shape[-1]
    P = as_f64(P)
    assert P.shape == (n + k + r, n + k + r) and tuple(M.shape) == (n + k + r, n + k)
    ctx.check(ctx.lib.edmdc_pinv_apply_dev(ctx.h, n, r, k, float(gamma), _dptr(C), int(nbags), int(L), int(x_bag_stride), int(u_bag_stride),
                                           _dptr(X), _dptr(U), _hptr(P), _dptr(M)), "edmdc_pinv_apply_dev")


def pinv_sym_device(G, ridge, rcond=1e-15):
    """pinv(G + ridge I) of the symmetric p x p Gram ON THE DEVICE: symmetric eigendecomposition (torch.linalg.eigh = the ROCm
    LAPACK library, a library primitive like a library GEMM) with numpy.linalg.pinv's cut-off -- eigenvalues (= singular values of
    a symmetric matrix, up to sign) not above rcond * the largest are dropped.  G: CUDA tensor [p, p]; returns a CUDA tensor.
    Opt-in (fit_dev(pinv="device")): 13 ms against 31-93 ms for the host's LAPACK pinv at p = 532 (tools/attic/time_pinv_device.py;
    the library's time is not monotone in p -- 87 ms at p = 520), P agrees with numpy's to 5e-12..2e-11 relative on the fixtures."""
    import torch
    A = G + ridge * torch.eye(G.shape[0], dtype=G.dtype, device=G.device)
    A = 0.5 * (A + A.T)
    w, Q = torch.linalg.eigh(A)
    cut = rcond * w.abs().max()
    winv = torch.where(w.abs() > cut, 1.0 / w, torch.zeros_like(w))
    return (Q * winv) @ Q.T


def pinv_sym_host(G, ridge, rcond=1e-15, safe=None):
    """pinv(G + ridge I) of the symmetric p x p Gram on the HOST through a symmetric eigendecomposition (LAPACK syevd) with
    numpy.linalg.pinv's cut-off, Q diag(1/w) Q^T: 13 ms instead of 29 ms at p = 520.  In exact arithmetic this is the reference's
    `pinv(...)` (Koopman/koopmanEDMDc.py:97,147); in floating point the two routes agree only while the matrix is comfortably
    conditioned: with the smallest computed eigenvalue above 1e-10 of the largest the scores agree to <= 1e-10, at 1e-12 they differ by
    1e-7 .. 1e-4, and at the class default ridge = 1e-8 with a wide kernel (cond ~ 1e14) the advisor of round 5 measured a 5 x worse
    training RMSE for this route than for numpy's SVD.  `safe`: return None instead of a matrix when the smallest eigenvalue is not above
    safe x the largest (pinv="auto" then calls numpy.linalg.pinv)."""
    A = G + ridge * np.eye(G.shape[0])
    w, Q = np.linalg.eigh(0.5 * (A + A.T))
    aw = np.abs(w)
    if safe is not None and not (w[0] > safe * aw.max()):
        return None
    keep = aw > rcond * aw.max()
    winv = np.zeros_like(w)
    winv[keep] = 1.0 / w[keep]
    return (Q * winv) @ Q.T


PINV_AUTO_SAFE = 1e-10        # pinv="auto": smallest / largest eigenvalue of G^T G + ridge I above which the eigendecomposition route is taken


def _chol_inverse(A):
    """(A^-1, kappa_1) of a symmetric positive definite A: a Cholesky factorisation as the test of definiteness, then LAPACK's inverse
    (numpy.linalg.inv; 5 ms at p = 520 where the eigendecomposition takes 13-15), symmetrised; kappa_1 = |A|_1 |A^-1|_1 -- for a symmetric
    matrix an UPPER bound of the 2-norm condition number.  None when A is not numerically positive definite.  NumPy only: SciPy's potri
    would halve the arithmetic, but importing scipy.linalg costs the first call of a process 0.1 s and starts a second, uncapped BLAS
    thread pool (measured: 2.8 s of CPU-quota throttling in the first fit())."""
    try:
        np.linalg.cholesky(A)
        Ai = np.linalg.inv(A)
    except np.linalg.LinAlgError:
        return None
    Ai = 0.5 * (Ai + Ai.T)
    return Ai, float(np.abs(A).sum(0).max() * np.abs(Ai).sum(0).max())


def _host_pinv_route(G, ridge, pinv):
    """(P, route) of the p x p solve of the normal equations on the host; route is "pinv", "cholesky" or "eigh".
    pinv="host": numpy.linalg.pinv, the reference's call (Koopman/koopmanEDMDc.py:97,147).
    pinv="auto" (default): the reference's own route wherever the route matters, something cheaper where it provably does not --
      1. the plain inverse (positive definite by Cholesky) when kappa_1(G^T G + ridge I) < 1 / PINV_AUTO_SAFE (then the smallest eigenvalue is above PINV_AUTO_SAFE x the
         largest, and the inverse IS the pseudo-inverse: nothing is cut off);
      2. otherwise the symmetric eigendecomposition with numpy.linalg.pinv's cut-off if the computed eigenvalue ratio is above the threshold;
      3. otherwise numpy.linalg.pinv.
    pinv="eigh": the eigendecomposition unconditionally (round 5's default; opt-in)."""
    if pinv == "host":
        return np.linalg.pinv(G + ridge * np.eye(G.shape[0])), "pinv"
    if pinv == "auto":
        A = G + ridge * np.eye(G.shape[0])
        A = 0.5 * (A + A.T)
        ci = _chol_inverse(A)
        if ci is not None and ci[1] * PINV_AUTO_SAFE < 1.0:
            return ci[0], "cholesky"
        hopeless = False
        try:                                                    # cond >= (max L_ii / min L_ii)^2: a lower bound that can rule the rest out
            dl = np.diag(np.linalg.cholesky(A))
            hopeless = (dl.min() / dl.max()) ** 2 < PINV_AUTO_SAFE
        except np.linalg.LinAlgError:
            hopeless = True
        P = None if hopeless else pinv_sym_host(G, ridge, safe=PINV_AUTO_SAFE)
        return (P, "eigh") if P is not None else (np.linalg.pinv(A), "pinv")
    if pinv in ("eigh", "device"):          # (the device form is fit_dev's; the host-list paths take the host's eigendecomposition)
        return pinv_sym_host(G, ridge), "eigh"
    raise ValueError("pinv must be 'auto', 'host', 'eigh' or 'device'")


def _host_pinv(G, ridge, pinv):
    return _host_pinv_route(G, ridge, pinv)[0]


def fit_dev(X, U, nbags, L, k, gamma, ridge, order="fit", centers=None, max_iter=300, tol=1e-4, random_state=0, ctx=None,
            timings=None, lift_cache=False, pinv="auto", bag_offsets=None):
    """The whole of KoopmanEDMDc.fit / fit_multi on device-resident data (DevArray or torch CUDA tensors -- the same launches, the same
    bits; bag layout: X [nbags*(L+1), n] states, U [nbags*L, r] inputs): centres with scikit-learn's KMeans stopping rule (Koopman/koopmanEDMDc.py:85: k-means++
    seeding, Lloyd up to max_iter 300, tol 1e-4) unless given, G^T[G|Y], the host pinv (:97/:147), and for order="fit" the
    two products of `(P G^T) Y` on the device (order="fit_multi": `P (G^T Y)` on the host).  Returns (A [d,d], B [d,r],
    centres [k,n] on the device); timings (dict) receives the stage wall times in seconds, the Lloyd iteration count and whether
    the stopping rule fired before max_iter.  lift_cache=True: keep the lifted rows of the Gram pass in HBM for the apply pass
    when they fit (edmdc_lift_cache; X, U, C are not touched in between): saves the second lift (11 ms per 1e7 pairs) for a
    45.7 GB block from torch's caching allocator -- whose FIRST allocation costs ~0.5 s (the driver hands out scrubbed memory),
    so it pays for repeated fits in one process, not for a single one; off by default.  pinv: how the p x p solve is done (_host_pinv) --
    "auto" (default): a symmetric eigendecomposition on the host when G^T G + ridge I is comfortably conditioned (13 instead of 29 ms at
    p = 520), numpy.linalg.pinv -- the reference's call, :97/:147 -- otherwise; "host": numpy.linalg.pinv always; "eigh": the
    eigendecomposition always (not the reference's scores once cond reaches ~1e12); "device": pinv_sym_device (torch.linalg.eigh:
    12 ms, first call in a process 0.2 s; torch tensors only).  bag_offsets (host int64 [nbags + 1]): a RAGGED trajectory list instead
    of nbags bags of L pairs -- X [rows, n] the stacked states, U [rows, r] row-aligned with X (upload_bags); nbags / L are ignored."""
    import time
    ctx = _ctx_of(X, ctx)
    ns = arrays_of(X, ctx)
    n, r = X.shape[-1], U.shape[-1]
    d, p = n + k, n + k + r
    tm = {} if timings is None else timings

    def tick():
        ns.sync()
        return time.perf_counter()

    t0 = tick()
    if centers is None:
        C, _, n_iter = kmeans_centers_dev(X.view(-1, n), k, random_state=random_state, max_iter=max_iter, tol=tol, ctx=ctx, timings=tm)
        tm["lloyd_iterations"], tm["lloyd_converged"] = n_iter, bool(n_iter < max_iter)
    else:
        C = centers
    t1 = tick()
    GG = ns.empty((p * p + p * d,))
    if ns.kind == "torch":
        GtG, GtY = GG[: p * p].view(p, p), GG[p * p:].view(p, d)
    else:
        GtG, GtY = GG.rows(0, p * p).view(p, p), GG.rows(p * p, p * p + p * d).view(p, d)
    if order == "fit":
        GtY = None                  # fit() never forms G^T Y (Koopman/koopmanEDMDc.py:89-97): its Gram pass is G^T G alone
    if order not in ("fit", "fit_multi"):
        raise ValueError("order must be 'fit' or 'fit_multi'")
    if pinv == "device" and ns.kind != "torch":
        raise ValueError("pinv='device' is torch.linalg.eigh: it needs torch tensors (arrays='torch')")
    cache_buf = None
    if order == "fit" and lift_cache:
        # keep the lifted rows of the Gram pass for the apply pass when HBM has room (rows x padded width x 8 B + slack)
        W = (k + 15) // 16 * 16 + (n + r + 15) // 16 * 16
        need = int(X.shape[0] * 1.01 + (1 << 21)) * (W + 1) * 8
        if ns.mem_free() > need + (4 << 30):
            # torch's caching allocator owns the block: a second fit() gets it back without a trip to the driver (a raw
            # hipMalloc of 45 GB right after a hipFree of the same size was seen to take 2.4 s)
            try:
                cache_buf = ns.empty((need,), np.uint8)
            except (RuntimeError, _lib.BrovError):
                cache_buf = None
    try:
        if cache_buf is not None:
            ctx.lift_cache(cache_buf.data_ptr(), need)
        if bag_offsets is not None:
            gram_ragged_dev(X, U, C, gamma, bag_offsets, GtG, GtY, ctx=ctx)
        else:
            gram_dev(X, U, C, gamma, nbags, L, L + 1, L, GtG, GtY, ctx=ctx)
        if getattr(ctx, "timing", False):
            tm["gram_kernel_ms"] = ctx.last_kernel_ms()
        if pinv == "device":
            t2 = tick()
            P = pinv_sym_device(GtG, ridge).cpu().numpy()
        elif pinv in ("auto", "host", "eigh"):
            Gh = ns.download(GtG)
            t2 = tick()
            with _blas_threads(p):
                P = _host_pinv(Gh, ridge, pinv)
        else:
            raise ValueError("pinv must be 'auto', 'host', 'eigh' or 'device'")
        t3 = time.perf_counter()
        if order == "fit":
            M = ns.empty((p, d))
            if bag_offsets is not None:
                pinv_apply_ragged_dev(X, U, C, gamma, bag_offsets, P, M, ctx=ctx)
            else:
                pinv_apply_dev(X, U, C, gamma, nbags, L, L + 1, L, P, M, ctx=ctx)
            Mt = ns.download(M).T
        else:
            with _blas_threads(p):
                Mt = (P @ ns.download(GtY)).T
    finally:
        if cache_buf is not None:              # withdraw the buffer before it goes back to the allocator, whatever happened
            ns.sync()
            ctx.lift_cache(None)
            del cache_buf
    t4 = tick()
    tm.update(centres_s=t1 - t0, gram_s=t2 - t1, pinv_s=t3 - t2, apply_s=t4 - t3, total_s=t4 - t0)
    return np.ascontiguousarray(Mt[:, :d]), np.ascontiguousarray(Mt[:, d:]), C


def usable_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a container that sees
    256 logical CPUs and is granted 16 is common on GPU hosts)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, int(q / per + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


class _blas_threads:
    """The host solve of the p x p normal matrix is the one BLAS/LAPACK call left on the fit path.  The reference caps its
    BLAS at 4 threads on import (Koopman/koopmanEDMDc.py:23-25); left alone, OpenBLAS starts one thread per VISIBLE CPU, and
    in a container that is granted fewer cores than it sees the solve takes 60-200 ms instead of 33 (p = 532).  Cap the pool for
    the duration of the solve: two threads up to p = 700 (round 5, tools/attic/time_eigh_threads.py on the GPU boxes' hosts: neither syevd
    nor gesdd at p ~ 520 gains from more -- eigh 18 / 14-17 / 17 / 18 ms, pinv 34 / 29-37 / 37 / 38 ms at 1 / 2 / 4 / 8 threads),
    the granted cores (at most 8) beyond; no-op when threadpoolctl is missing."""
    _cores = None

    def __init__(self, p=0):
        self.p = int(p)

    def __enter__(self):
        self._ctx = None
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:
            return self
        if _blas_threads._cores is None:
            _blas_threads._cores = usable_cores()
        limit = min(2 if self.p <= 700 else 8, _blas_threads._cores)
        self._ctx = threadpool_limits(limits=limit, user_api="blas")
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def solve_AB_fit_order(X_list, U_list, C, gamma, GtG, ridge, d, ctx=None, pinv="auto"):
    """(A, B) exactly as KoopmanEDMDc.fit associates the product (Koopman/koopmanEDMDc.py:97-101):
    M = (pinv(G^T G + ridge I) @ G.T) @ Y, the pinv on the host (numpy, like the reference), the two large products on
    the GPU.  Better conditioned than fit_multi's pinv(.) @ (G^T Y): at the class defaults (k = 200, ridge = 1e-8) the two
    differ by 1e-6 in the H = 100 RMSE."""
    with _blas_threads(GtG.shape[0]):
        P = _host_pinv(GtG, ridge, pinv)
    M = pinv_apply(X_list, U_list, C, gamma, P, ctx=ctx).T
    return np.ascontiguousarray(M[:, :d]), np.ascontiguousarray(M[:, d:])


def solve_AB(GtG, GtY, ridge, d, pinv="auto"):
    """Host solve of the ridge normal equations exactly as the reference does it
    (Koopman/koopmanEDMDc.py:147-151): M = pinv(G^T G + ridge I) (G^T Y); A = M^T[:, :d]; B = M^T[:, d:]."""
    with _blas_threads(GtG.shape[0]):
        M = _host_pinv(GtG, ridge, pinv) @ GtY
    M = M.T
    return np.ascontiguousarray(M[:, :d]), np.ascontiguousarray(M[:, d:])


def multistep_se(X, U, C, gamma, A, B, H, want_xhat=False, ctx=None):
    """H-step lifted propagation + endpoint squared error (KoopmanEDMDc.multistep_rmse / evaluate)."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    X = as_f64(X)
    U = as_f64(U)
    C = as_f64(C)
    A = as_f64(A)
    B = as_f64(B)
    N, n = X.shape
    k, r = C.shape[0], U.shape[1]
    ns = N - H
    # like the reference (Koopman/koopmanEDMDc.py:172-200) accept len(U) == len(X) - 1: only rows 0..N-2 are ever read
    assert U.shape[0] >= N - 1, f"U has {U.shape[0]} rows, need at least len(X) - 1 = {N - 1}"
    xhat = np.empty((max(ns, 0), n)) if want_xhat else None
    se = ctypes.c_double(0.0)
    ctx.check(ctx.lib.edmdc_multistep_se(ctx.h, n, r, k, float(gamma), _hptr(C), _hptr(A), _hptr(B), N, int(H), _hptr(X), _hptr(U),
                                         ctypes.addressof(se), _hptr(xhat)), "edmdc_multistep_se")
    return se.value, xhat


def linear_coefficients(A, B, n, H):
    """(RHt [d, n], Gt [H, r, n]) of multistep_se_linear: RHt = (E A^H)^T, Gt[t] = (E A^(H-1-t) B)^T with E the first n rows of the identity --
    H products of an n x d block by A on the host (NumPy)."""
    A, B = as_f64(A), as_f64(B)
    d, r = A.shape[0], B.shape[1]
    R = np.zeros((n, d))
    R[:, :n] = np.eye(n)
    Gt = np.empty((H, r, n))
    for j in range(H):                  # R = E A^j
        Gt[H - 1 - j] = (R @ B).T
        R = R @ A
    return np.ascontiguousarray(R.T), Gt


def multistep_se_linear(X, U, C, gamma, A, B, H, want_xhat=False, ctx=None):
    """multistep_se by linearity (opt-in): x_hat[w] = (E A^H) phi(x_w) + sum_t (E A^(H-1-t) B) u_{w+t} in ONE pass over the windows
    (edmdc_multistep_se_linear) instead of H lifted GEMM steps.  Same arguments and results as multistep_se up to the rounding of the
    explicit powers of A."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    X, U, C = as_f64(X), as_f64(U), as_f64(C)
    N, n = X.shape
    k, r = C.shape[0], U.shape[1]
    ns = N - H
    assert U.shape[0] >= N - 1, f"U has {U.shape[0]} rows, need at least len(X) - 1 = {N - 1}"
    RHt, Gt = linear_coefficients(A, B, n, int(H))
    xhat = np.empty((max(ns, 0), n)) if want_xhat else None
    se = ctypes.c_double(0.0)
    ctx.check(ctx.lib.edmdc_multistep_se_linear(ctx.h, n, r, k, float(gamma), _hptr(C), _hptr(RHt), _hptr(Gt) if H else None, N, int(H),
                                                _hptr(X), _hptr(U), ctypes.addressof(se), _hptr(xhat)), "edmdc_multistep_se_linear")
    return se.value, xhat


def simulate_lifted(x0, U_seq, C, gamma, A, B, ctx=None):
    """KoopmanEDMDc.simulate, batched: x0 [nb,n], U_seq [nb,T,r] -> [nb,T+1,n]."""
    ctx = ctx or default_context()
    ctx.use_null_stream()
    x0 = as_f64(x0)
    U_seq = as_f64(U_seq)
    nb, n = x0.shape
    T, r = U_seq.shape[1], U_seq.shape[2]
    C = as_f64(C)
    A = as_f64(A)                            # named references: np.load can hand back Fortran-ordered arrays,
    B = as_f64(B)                            # as_f64 then copies, and only the address crosses the ABI
    out = np.empty((nb, T + 1, n))
    ctx.check(ctx.lib.edmdc_simulate(ctx.h, n, r, C.shape[0], float(gamma), _hptr(C), _hptr(A), _hptr(B), nb, T,
                                     _hptr(x0), _hptr(U_seq), _hptr(out)), "edmdc_simulate")
    return out
