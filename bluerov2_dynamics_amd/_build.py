"""Build recipe of libbrov2.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.environ.get("BROV2_LIBRARY") or os.path.join(PKG, "libbrov2.so")     # override: A/B runs of experimental builds
SOURCES = ["capi.hip", "rollout.hip", "edmdc.hip", "propagate.hip", "kmeans.hip", "controls.hip", "comm.hip"]
HEADERS = ["brov2_device.h", "brov2_fast.h", "brov2_kernels.h", os.path.join("..", "..", "include", "brov2.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-DBROV2_BUILDING=1"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libbrov2.so can only be built with the ROCm toolchain")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip into bluerov2_dynamics_amd/libbrov2.so (cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    extra = os.environ.get("BROV2_HIPCC_EXTRA", "").split()      # experiments only (e.g. -DBROV_STAGE_RELOAD=0)
    cmd = [hipcc()] + FLAGS + extra + ["-o", LIB + ".tmp"] + srcs + ["-ldl"]      # dl: librccl is bound at run time (comm.hip)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
