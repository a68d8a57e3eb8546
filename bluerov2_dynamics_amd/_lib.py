"""ctypes binding of libbrov2.so (include/brov2.h).  There is NO CPU fallback: if the HIP
library or a gfx950 device is missing, construction of a Context raises."""
import ctypes
import os
import threading

import numpy as np

from . import _build

c_double_p = ctypes.POINTER(ctypes.c_double)
c_void_p = ctypes.c_void_p
i64 = ctypes.c_int64

# enums of include/brov2.h
THRUSTER_EULER, WRENCH_EULER, WRENCH_QUAT = 0, 1, 2
DI_THRUSTER_EULER, DI_WRENCH_EULER, DI_WRENCH_QUAT = 3, 4, 5
EULER, RK4 = 0, 1
LAG_PER_CALL, LAG_PER_STEP = 0, 1
LAYOUT_BTU, LAYOUT_TUB, LAYOUT_TPB = 0, 1, 2
DIST_IID_UNIFORM, DIST_AR1 = 0, 1
NX = {THRUSTER_EULER: 12, WRENCH_EULER: 12, WRENCH_QUAT: 13, DI_THRUSTER_EULER: 12, DI_WRENCH_EULER: 12, DI_WRENCH_QUAT: 13}
NU = {THRUSTER_EULER: 8, WRENCH_EULER: 6, WRENCH_QUAT: 6, DI_THRUSTER_EULER: 8, DI_WRENCH_EULER: 6, DI_WRENCH_QUAT: 6}
STATUS = {0: "BROV_OK", -1: "BROV_ERR_ARG", -2: "BROV_ERR_HIP", -3: "BROV_ERR_NOMEM", -4: "BROV_ERR_NODEVICE", -5: "BROV_ERR_COMM"}
COMM_ID_BYTES = 128


class BrovParams(ctypes.Structure):
    """struct brov_params (include/brov2.h)."""
    _fields_ = [
        ("rho", ctypes.c_double), ("g", ctypes.c_double), ("m", ctypes.c_double), ("volume", ctypes.c_double),
        ("xb", ctypes.c_double), ("yb", ctypes.c_double), ("zb", ctypes.c_double),
        ("Ix", ctypes.c_double), ("Iy", ctypes.c_double), ("Iz", ctypes.c_double),
        ("added_mass", ctypes.c_double * 6), ("lin_damp", ctypes.c_double * 6), ("quad_damp", ctypes.c_double * 6),
        ("current", ctypes.c_double * 3),
        ("thr_r", (ctypes.c_double * 3) * 8), ("thr_dir", (ctypes.c_double * 3) * 8),
        ("thrust_poly", ctypes.c_double * 5),
        ("lag_Ac", ctypes.c_double * 9), ("lag_Bc", ctypes.c_double * 3), ("lag_Cc", ctypes.c_double * 3),
    ]


class BrovError(RuntimeError):
    pass


# callback types of include/brov2.h
FAR_SELECT_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, c_double_p, i64, ctypes.c_int, ctypes.POINTER(i64))
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, c_void_p, i64, ctypes.c_int)


def _numpy_far_select(_user, dist_p, N, n_empty, out_p):
    """The rows scikit-learn relocates its empty clusters to: `np.argpartition(distances, -n_empty)[:-n_empty-1:-1]`
    (sklearn/cluster/_k_means_common.pyx, `_relocate_empty_clusters_dense`) -- NumPy's own introselect on this host, so that
    ties and the order of the n_empty farthest rows fall exactly as they do inside scikit-learn."""
    try:
        d = np.ctypeslib.as_array(dist_p, shape=(int(N),))
        far = np.argpartition(d, -int(n_empty))[:-int(n_empty) - 1:-1]
        out = np.ctypeslib.as_array(out_p, shape=(int(n_empty),))
        out[:] = far
        return 0
    except Exception:                      # never let an exception cross the C boundary
        return 1


_NUMPY_FAR_SELECT = FAR_SELECT_FN(_numpy_far_select)      # one process-wide trampoline (must outlive every ctx that holds it)


_lib = None
_lib_lock = threading.Lock()

# name -> (restype, argtypes); every symbol declared in include/brov2.h
SIGNATURES = {
    "brov_abi_version": (ctypes.c_int, []),
    "brov_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_void_p)]),
    "brov_destroy": (None, [c_void_p]),
    "brov_last_error": (ctypes.c_char_p, [c_void_p]),
    "brov_arch_is_supported": (ctypes.c_int, [ctypes.c_char_p]),
    "brov_device_arch": (ctypes.c_int, [c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    "brov_xcd_round_robin": (ctypes.c_int, [c_void_p]),
    "brov_set_stream": (ctypes.c_int, [c_void_p, c_void_p]),
    "brov_sync": (ctypes.c_int, [c_void_p]),
    "brov_set_timing": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "brov_last_kernel_ms": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "brov_default_params": (None, [ctypes.POINTER(BrovParams)]),
    "brov_set_params": (ctypes.c_int, [c_void_p, ctypes.POINTER(BrovParams)]),
    "brov_get_params": (ctypes.c_int, [c_void_p, ctypes.POINTER(BrovParams)]),
    "brov_get_derived": (ctypes.c_int, [ctypes.POINTER(BrovParams), c_double_p, c_double_p]),
    "brov_discretise_lag": (ctypes.c_int, [ctypes.POINTER(BrovParams), ctypes.c_double, c_double_p, c_double_p]),
    "brov_model_nx": (ctypes.c_int, [ctypes.c_int]),
    "brov_model_nu": (ctypes.c_int, [ctypes.c_int]),
    "brov_malloc": (ctypes.c_int, [c_void_p, ctypes.c_size_t, ctypes.POINTER(c_void_p)]),
    "brov_free": (ctypes.c_int, [c_void_p, c_void_p]),
    "brov_memcpy_h2d": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]),
    "brov_memcpy_d2h": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]),
    "brov_memset": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, ctypes.c_size_t]),
    "brov_mem_info": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "edmdc_col_stats_dev": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, i64, c_void_p, c_void_p]),
    "brov_rhs": (ctypes.c_int, [c_void_p, ctypes.c_int, i64, c_void_p, c_void_p, ctypes.c_double, c_void_p, c_void_p]),
    "brov_thruster_forces": (ctypes.c_int, [c_void_p, i64, c_void_p, ctypes.c_double, c_void_p, c_void_p]),
    "brov_rollout": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, i64,
                                    ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, i64, c_void_p]),
    "brov_set_btu_staging": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "brov_set_di_gains": (ctypes.c_int, [c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "brov_rollout_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, i64,
                                        ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, i64, c_void_p]),
    "brov_window_endpoint_se": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, i64, i64, ctypes.c_double,
                                               c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "brov_window_endpoint_se_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, i64, i64, ctypes.c_double,
                                                   c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "brov_fill_controls_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, i64, i64, ctypes.c_int,
                                              ctypes.c_uint64, i64, i64, c_void_p, c_void_p]),
    "edmdc_lift": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
    "edmdc_gram": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                  i64, i64, i64, i64, c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "edmdc_gram_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                      i64, i64, i64, i64, c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "edmdc_gram_ragged": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                         i64, c_void_p, c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "edmdc_gram_ragged_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                             i64, c_void_p, c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "edmdc_pinv_apply_ragged": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                               i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edmdc_pinv_apply_ragged_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                                   i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "brov_upload_bags": (ctypes.c_int, [c_void_p, i64, c_void_p, c_void_p, c_void_p, ctypes.c_int, c_void_p]),
    "edmdc_set_chunk_rows": (ctypes.c_int, [c_void_p, i64]),
    "edmdc_kmeans_lloyd": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_void_p, ctypes.c_int,
                                          ctypes.c_double, c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "edmdc_kmeans_lloyd_dev": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, ctypes.c_int, c_void_p, i64, c_void_p, c_void_p, ctypes.c_int,
                                              ctypes.c_double, c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "edmdc_kmeanspp_dev": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, ctypes.c_int, c_void_p, i64, c_void_p, i64, ctypes.c_int, c_void_p,
                                          c_void_p, c_void_p]),
    "edmdc_multistep_se": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                          c_void_p, c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edmdc_multistep_se_linear": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                                 c_void_p, c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edmdc_simulate": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                      c_void_p, c_void_p, i64, i64, c_void_p, c_void_p, c_void_p]),
    "edmdc_pinv_apply": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                        i64, i64, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edmdc_pinv_apply_dev": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void_p,
                                            i64, i64, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "edmdc_gram_decomposition": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "edmdc_gtg_decomposition": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "edmdc_apply_decomposition": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "edmdc_set_apply_variant": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "edmdc_lift_cache": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "edmdc_set_kmeans_variant": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "edmdc_set_kmeans_bounds_rate": (ctypes.c_int, [c_void_p, ctypes.c_double]),
    "brov_experiments_build": (ctypes.c_int, []),
    "brov_set_rollout_variant": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "edmdc_set_kmeans_far_select": (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    "edmdc_kmeans_relocations": (ctypes.c_int, [c_void_p]),
    "edmdc_kmeans_loop_info": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "edmdc_far_select_numpy": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p]),
    "edmdc_set_kmeans_allreduce": (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    "edmdc_set_kmeans_shard": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, i64, i64]),
    "brov_comm_available": (ctypes.c_int, []),
    "brov_comm_unique_id": (ctypes.c_int, [c_void_p]),
    "brov_comm_init_rank": (ctypes.c_int, [ctypes.c_int, c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_void_p)]),
    "brov_comm_destroy": (None, [c_void_p]),
    "brov_comm_nranks": (ctypes.c_int, [c_void_p]),
    "brov_comm_rank": (ctypes.c_int, [c_void_p]),
    "brov_comm_last_error": (ctypes.c_char_p, [c_void_p]),
    "edmdc_gram_allreduce_dev": (ctypes.c_int, [c_void_p, c_void_p, i64, c_void_p, i64, c_void_p]),
    "brov_comm_allreduce_words": (ctypes.c_int, [c_void_p, c_void_p, i64, ctypes.c_int, c_void_p]),
    "edmdc_kmeans_use_comm": (ctypes.c_int, [c_void_p, c_void_p]),
}


def library_path():
    return _build.LIB


hip_runtime = None        # how load_library() settled the HIP runtime question (diagnostics; see _one_hip_runtime)


def _one_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7, like /opt/rocm's) and links
    it by the unversioned name: if libbrov2.so has pulled in /opt/rocm's copy first, a later `import torch` loads its own as a SECOND
    runtime and finds no GPU.  The drop-in classes themselves need no torch (north_star: torch only on the PINc path), and importing it
    costs ~0.8 s of a script's first second, so by default torch is NOT imported here:
      * torch already imported by the caller (the reference's PINc scripts import it at the top): nothing to do, its runtime serves both;
      * BROV2_TORCH=auto (default), torch installed but not imported: only torch's libamdhip64.so is loaded (the very file torch would
        load -- the loader recognises it by inode), so a later `import torch` still shares the runtime.  Caveat, measured with
        tools/attic/time_late_torch.py: once the runtime has been INITIALISED (a Context exists), HIP registers torch's code objects eagerly and
        that later import takes ~10 s instead of ~0.8 s -- a script that wants torch should import it before its first use of this package;
      * BROV2_TORCH=1: import torch here, first (the behaviour up to round 5);
      * BROV2_TORCH=0: never look for torch; libbrov2.so binds /opt/rocm's runtime (a later `import torch` in the same process would then
        not see the GPU)."""
    import sys
    if "torch" in sys.modules:
        return "torch (imported by the caller)"
    mode = os.environ.get("BROV2_TORCH", "auto").strip().lower()
    if mode in ("1", "import", "yes", "true"):
        try:
            import torch  # noqa: F401
            return "torch (imported by load_library: BROV2_TORCH=1)"
        except ImportError:
            return "system (BROV2_TORCH=1 but torch is not installed)"
    if mode in ("0", "no", "never", "false"):
        return "system (BROV2_TORCH=0)"
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")          # locates the package; imports nothing
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
                return "torch's libamdhip64.so preloaded (torch itself not imported)"
            except OSError:
                pass
    return "system"


def load_library():
    """dlopen libbrov2.so (needs libamdhip64; no GPU needed just to load) and bind every symbol."""
    global _lib, hip_runtime
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise BrovError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        hip_runtime = _one_hip_runtime()
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.brov_abi_version() != 1:
            raise BrovError("libbrov2.so ABI version mismatch")
        _lib = lib
        return lib


def default_params() -> BrovParams:
    p = BrovParams()
    load_library().brov_default_params(ctypes.byref(p))
    return p


def discretise_lag(dt: float, params: BrovParams = None):
    """ZOH (Ad, Bd) of the thruster lag -- host only (fossen/BlueROV2.py:490-496)."""
    Ad = np.zeros((3, 3))
    Bd = np.zeros(3)
    rc = load_library().brov_discretise_lag(ctypes.byref(params) if params is not None else None, float(dt),
                                            Ad.ctypes.data_as(c_double_p), Bd.ctypes.data_as(c_double_p))
    if rc != 0:
        raise BrovError(f"brov_discretise_lag failed: {STATUS.get(rc, rc)}")
    return Ad, Bd


def derived(params: BrovParams = None):
    """(Minv diagonal [6], allocation matrix [6,8]) of a parameter set -- host only."""
    Minv = np.zeros(6)
    T = np.zeros((6, 8))
    rc = load_library().brov_get_derived(ctypes.byref(params) if params is not None else None,
                                         Minv.ctypes.data_as(c_double_p), T.ctypes.data_as(c_double_p))
    if rc != 0:
        raise BrovError(f"brov_get_derived failed: {STATUS.get(rc, rc)}")
    return Minv, T


def _hptr(a):
    """host pointer of a C-contiguous float64 ndarray (or None)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One brov_ctx = one device + one stream.  Not thread-safe (like the reference objects)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = c_void_p()
        rc = self.lib.brov_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise BrovError(f"brov_create(device={device}) failed: {STATUS.get(rc, rc)} -- a gfx950 GPU is required, "
                            "there is no CPU fallback")
        self.h = h
        self.device = int(device)
        self._stream = 0          # handle the ctx currently launches on (0 = the null stream)
        self._allreduce_cb = None
        # empty clusters of the Lloyd loop are relocated to the rows NumPy's argpartition picks, as in scikit-learn
        self.check(self.lib.edmdc_set_kmeans_far_select(self.h, ctypes.cast(_NUMPY_FAR_SELECT, c_void_p), None), "edmdc_set_kmeans_far_select")

    @property
    def arch(self) -> str:
        buf = ctypes.create_string_buffer(64)
        self.check(self.lib.brov_device_arch(self.h, buf, 64), "brov_device_arch")
        return buf.value.decode()

    @property
    def xcd_round_robin(self) -> int:
        """1 if blocks b and b+8 shared an XCD when the ctx was created (the kernels' L2-sharing assumption), 0 / -1 otherwise."""
        return int(self.lib.brov_xcd_round_robin(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.brov_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.lib.brov_last_error(self.h)
            raise BrovError(f"{what}: {STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    # -- parameters
    def get_params(self) -> BrovParams:
        p = BrovParams()
        self.check(self.lib.brov_get_params(self.h, ctypes.byref(p)), "brov_get_params")
        return p

    def set_params(self, p: BrovParams):
        self.check(self.lib.brov_set_params(self.h, ctypes.byref(p)), "brov_set_params")

    # -- stream / timing
    def set_stream(self, stream_handle):
        """Launch on another HIP stream.  The library orders the new stream behind whatever this ctx still has queued on
        the old one (its scratch buffers are shared), so switching is safe at any point."""
        stream_handle = int(stream_handle or 0)
        if stream_handle != self._stream:
            self.check(self.lib.brov_set_stream(self.h, c_void_p(stream_handle)), "brov_set_stream")
            self._stream = stream_handle

    def use_torch_stream(self):
        """Device path: launch on torch's current stream for this device."""
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def use_null_stream(self):
        """Host path (arrays in, arrays out, synchronous): back to the null stream, so that a torch stream handle bound by
        an earlier device-path call -- possibly destroyed since -- is never used again."""
        self.set_stream(0)

    def set_di_gains(self, K_lin, K_ang):
        """Gains [nu,3] of the double-integrator models (include/brov2.h: brov_set_di_gains)."""
        K_lin, K_ang = as_f64(K_lin), as_f64(K_ang)
        assert K_lin.shape == K_ang.shape and K_lin.shape[1] == 3
        self.check(self.lib.brov_set_di_gains(self.h, int(K_lin.shape[0]), K_lin.ctypes.data, K_ang.ctypes.data), "brov_set_di_gains")

    def set_btu_staging(self, mode: int):
        """0 auto, 1 always stage BTU tiles through LDS, 2 never (see include/brov2.h)."""
        self.check(self.lib.brov_set_btu_staging(self.h, int(mode)), "brov_set_btu_staging")

    def lift_cache(self, device_ptr, nbytes: int = 0):
        """Lend the ctx a device buffer for the lifted rows of the fit() sequence gram -> apply (include/brov2.h:
        edmdc_lift_cache); device_ptr None / 0 withdraws it."""
        self.check(self.lib.edmdc_lift_cache(self.h, c_void_p(device_ptr or 0), int(nbytes if device_ptr else 0)), "edmdc_lift_cache")

    def set_kmeans_variant(self, variant: int):
        """Lloyd's loop (include/brov2.h: edmdc_set_kmeans_variant): 0 = default (candidate filter on sorted samples, packed-fp32
        screening, distance bounds), + 1 = full scan, + 2 = filter without the sorted order, + 4 = distance bounds off -- all with the
        same labels and centres.  (An experiments build accepts more bits: KMV_* in csrc/capi.hip.)"""
        self.check(self.lib.edmdc_set_kmeans_variant(self.h, int(variant)), "edmdc_set_kmeans_variant")

    def set_kmeans_bounds_rate(self, rate: float):
        """Share of labels changed per iteration below which the E-steps walk the list of failed bounds only (default 0.03; speed only)."""
        self.check(self.lib.edmdc_set_kmeans_bounds_rate(self.h, float(rate)), "edmdc_set_kmeans_bounds_rate")

    def set_rollout_variant(self, variant: int):
        """Thruster-model rollouts: 0 = two-wave kernel (default), 1 = one-lane kernel (second implementation)."""
        self.check(self.lib.brov_set_rollout_variant(self.h, int(variant)), "brov_set_rollout_variant")

    def set_kmeans_far_select(self, numpy_rule: bool):
        """True (default): the empty-cluster relocation picks its rows by calling np.argpartition on this host, like scikit-learn here;
        False: the library's own rule -- a restatement of NumPy's (non-SIMD) introselect, what a plain-C caller gets
        (include/brov2.h: edmdc_set_kmeans_far_select, edmdc_far_select_numpy).  Sharded runs apply whichever is set to the
        distances of all ranks' rows."""
        fn = ctypes.cast(_NUMPY_FAR_SELECT, c_void_p) if numpy_rule else None
        self.check(self.lib.edmdc_set_kmeans_far_select(self.h, fn, None), "edmdc_set_kmeans_far_select")

    def kmeans_relocations(self) -> int:
        """Iterations of the last Lloyd call in which empty clusters were relocated."""
        return int(self.lib.edmdc_kmeans_relocations(self.h))

    def kmeans_loop_info(self) -> dict:
        """How the last Lloyd call ran (include/brov2.h: edmdc_kmeans_loop_info)."""
        v = (ctypes.c_int * 4)()
        self.check(self.lib.edmdc_kmeans_loop_info(self.h, v), "edmdc_kmeans_loop_info")
        return dict(relocations=v[0], resorts=v[1], first_resort_iteration=v[2], list_form_e_steps=v[3])

    def set_kmeans_allreduce(self, fn):
        """Sharded Lloyd (include/brov2.h: edmdc_set_kmeans_allreduce).  fn(device_ptr: int, count: int, op: int) combines the
        device buffer over all ranks in place (op 0: sum of int64, op 1: max of uint64), ordered with respect to the ctx stream;
        None switches back to a single rank."""
        if fn is None:
            self._allreduce_cb = None
            self.check(self.lib.edmdc_set_kmeans_allreduce(self.h, None, None), "edmdc_set_kmeans_allreduce")
            return

        def tramp(_user, ptr, count, op):
            try:
                fn(int(ptr), int(count), int(op))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._allreduce_cb = ALLREDUCE_FN(tramp)
        self.check(self.lib.edmdc_set_kmeans_allreduce(self.h, ctypes.cast(self._allreduce_cb, c_void_p), None), "edmdc_set_kmeans_allreduce")

    def set_kmeans_shard(self, rank: int = 0, world: int = 1, row_offset: int = 0, n_global: int = 0):
        """This rank's place in a sharded k-means (include/brov2.h: edmdc_set_kmeans_shard); the defaults are a single rank."""
        self.check(self.lib.edmdc_set_kmeans_shard(self.h, int(rank), int(world), int(row_offset), int(n_global)), "edmdc_set_kmeans_shard")

    def kmeans_use_comm(self, comm):
        """Sharded Lloyd through the torch-free communicator (a _lib.Comm, or None for a single rank again)."""
        self.check(self.lib.edmdc_kmeans_use_comm(self.h, comm.h if comm is not None else None), "edmdc_kmeans_use_comm")

    def set_apply_variant(self, variant: int):
        """edmdc_pinv_apply: 0 = tuned W-rows kernel, 1 = the plain second implementation (see include/brov2.h)."""
        self.check(self.lib.edmdc_set_apply_variant(self.h, int(variant)), "edmdc_set_apply_variant")

    def sync(self):
        self.check(self.lib.brov_sync(self.h), "brov_sync")

    def set_timing(self, on: bool):
        self.check(self.lib.brov_set_timing(self.h, int(bool(on))), "brov_set_timing")
        self.timing = bool(on)

    def last_kernel_ms(self) -> float:
        ms = ctypes.c_float(0.0)
        self.check(self.lib.brov_last_kernel_ms(self.h, ctypes.byref(ms)), "brov_last_kernel_ms")
        return float(ms.value)


_default = {}
_default_lock = threading.Lock()


def default_context(device: int = None) -> Context:
    """Process-wide context per device (device defaults to $BROV2_DEVICE, then $LOCAL_RANK, then 0)."""
    if device is None:
        device = int(os.environ.get("BROV2_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _default.get(device)
    if ctx is None:
        with _default_lock:                      # (a warm_up() thread may be creating it right now)
            ctx = _default.get(device)
            if ctx is None:
                ctx = _default[device] = Context(device)
    return ctx


def warm_up(device: int = None, block: bool = False):
    """Start creating the process-wide context NOW, in a background thread: binding the library and initialising the HIP runtime is
    0.15-0.2 s of the first call of a fresh process (profiles/r06_fit_time.txt) and needs nothing from the caller -- a script that calls
    this right after its imports overlaps it with its own start-up (pandas import, CSV parsing; the reference's scripts spend
    seconds there before their first fit()).  The ctypes calls release the GIL.  Errors (no GPU) are left for the first real use to
    raise.  Opt-in: nothing happens at import time.  block=True waits for the context (returns it)."""
    def work():
        try:
            default_context(device)
        except Exception:                        # reported by whoever needs the context
            pass
    if block:
        return default_context(device)
    t = threading.Thread(target=work, name="brov2-warm-up", daemon=True)
    t.start()
    return t


class Comm:
    """One RCCL communicator (brov_comm): the direct, torch-free form of the path's only collective.  Rank 0 creates the
    id with Comm.unique_id() and hands the 128 bytes to the other ranks; every rank then constructs Comm(device, id,
    nranks, rank) (collective)."""

    def __init__(self, device: int, unique_id: bytes, nranks: int, rank: int):
        self.lib = load_library()
        assert len(unique_id) == COMM_ID_BYTES
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = c_void_p()
        rc = self.lib.brov_comm_init_rank(int(device), ctypes.addressof(buf), int(nranks), int(rank), ctypes.byref(h))
        if rc != 0:
            raise BrovError(f"brov_comm_init_rank: {STATUS.get(rc, rc)}: {self.lib.brov_comm_last_error(None).decode()}")
        self.h = h
        self.nranks, self.rank, self.device = int(nranks), int(rank), int(device)

    @staticmethod
    def available() -> bool:
        return bool(load_library().brov_comm_available())

    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
        rc = lib.brov_comm_unique_id(ctypes.addressof(buf))
        if rc != 0:
            raise BrovError(f"brov_comm_unique_id: {STATUS.get(rc, rc)}: {lib.brov_comm_last_error(None).decode()}")
        return bytes(buf)

    def allreduce_gram_(self, GtG, GtY, stream_handle=None):
        """In-place sum over ranks of two fp64 device arrays, one grouped RCCL call: torch CUDA tensors (on torch's current stream) or
        engine.DevArray (on the null stream their ctx launches on) -- the torch-free form of the path's only collective."""
        if type(GtG).__module__.startswith("torch"):
            import torch
            assert GtG.is_cuda and GtY.is_cuda and GtG.dtype == torch.float64 and GtY.dtype == torch.float64
            assert GtG.is_contiguous() and GtY.is_contiguous()
            st = torch.cuda.current_stream(GtG.device).cuda_stream if stream_handle is None else stream_handle
        else:
            assert GtG.dtype == np.float64 and GtY.dtype == np.float64
            st = 0 if stream_handle is None else stream_handle
        rc = self.lib.edmdc_gram_allreduce_dev(self.h, GtG.data_ptr(), GtG.numel(), GtY.data_ptr(), GtY.numel(), c_void_p(st or 0))
        if rc != 0:
            raise BrovError(f"edmdc_gram_allreduce_dev: {STATUS.get(rc, rc)}: {self.lib.brov_comm_last_error(self.h).decode()}")
        return GtG, GtY

    def close(self):
        if getattr(self, "h", None):
            self.lib.brov_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
