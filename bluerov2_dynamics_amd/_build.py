"""Build recipe of libbrov2.so (hipcc, gfx950 only, in-tree so the .so travels with the repo).

One object per source file under csrc/build/ (compiled in parallel, rebuilt only when the file or a header changed),
linked into bluerov2_dynamics_amd/libbrov2.so.  `variant(...)` builds an experimental copy of the library with extra
compiler flags on some files (A/B runs of kernel experiments; selected at run time with $BROV2_LIBRARY)."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.environ.get("BROV2_LIBRARY") or os.path.join(PKG, "libbrov2.so")     # override: A/B runs of experimental builds
SOURCES = ["capi.hip", "rollout.hip", "edmdc.hip", "propagate.hip", "kmeans.hip", "sortperm.hip", "controls.hip", "comm.hip", "colstats.hip"]
HEADERS = ["brov2_device.h", "brov2_fast.h", "brov2_kernels.h", os.path.join("..", "..", "include", "brov2.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DBROV2_BUILDING=1"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]      # dl: librccl is bound at run time (comm.hip)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libbrov2.so can only be built with the ROCm toolchain")
    return exe


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))


def stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src, obj, extra, verbose):
    cmd = [hipcc()] + CFLAGS + extra + ["-c", src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(obj + ".tmp", obj)


def _objects(objdir, force, verbose, extra_for=None):
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _newest_header()
    env_extra = os.environ.get("BROV2_HIPCC_EXTRA", "").split()      # experiments only
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = env_extra + list((extra_for or {}).get(s, []))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((src, obj, extra))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    return objs


def _link(objs, lib, verbose):
    cmd = [hipcc()] + objs + LDFLAGS + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(lib + ".tmp", lib)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip into bluerov2_dynamics_amd/libbrov2.so (cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    objs = _objects(os.path.join(CSRC, "build"), force, verbose)
    _link(objs, LIB, verbose)
    return LIB


def variant(name, flags_for, verbose=False):
    """Experimental build: build_variants/<name>/libbrov2.so with `flags_for[file]` appended to that file's compile line
    (the other objects are shared with the main build).  Returns the library path (use it as $BROV2_LIBRARY)."""
    root = os.path.join(os.path.dirname(PKG), "build_variants", name)
    os.makedirs(root, exist_ok=True)
    base = dict(zip(SOURCES, _objects(os.path.join(CSRC, "build"), False, verbose)))
    objs = []
    for s in SOURCES:
        if s in flags_for:
            obj = os.path.join(root, s.replace(".hip", ".o"))
            _compile(os.path.join(CSRC, s), obj, list(flags_for[s]), verbose)
            objs.append(obj)
        else:
            objs.append(base[s])
    lib = os.path.join(root, "libbrov2.so")
    _link(objs, lib, verbose)
    return lib


def bagtable_path():
    import sysconfig
    return os.path.join(PKG, "_bagtable" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_bagtable(force=False, verbose=False):
    """gcc the CPython helper csrc/bagtable.c (the header walk of fit_multi's trajectory lists; no arithmetic) into the package
    directory, next to libbrov2.so.  Optional at run time: engine.BagTable falls back to its Python loop without it."""
    import sysconfig
    src, out = os.path.join(CSRC, "bagtable.c"), bagtable_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler for the _bagtable helper")
    cmd = [cc, "-O2", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"], src, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_bagtable(force=True, verbose=True))
