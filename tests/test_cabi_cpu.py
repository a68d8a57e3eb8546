"""CPU-only checks of the drop-in boundary: libbrov2.so loads, exports every symbol that
include/brov2.h declares, its host-only entry points match the reference fixtures, and the product
never routes through the oracle / a CPU fallback.  No compute kernels are called."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden, rel_err


@pytest.fixture(scope="module")
def lib():
    from bluerov2_dynamics_amd import _build, _lib
    _build.build_library()          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load_library()


def header_symbols():
    txt = open(os.path.join(REPO, "include", "brov2.h")).read()
    return re.findall(r"^BROV_API\s+[\w\s\*]+?\b((?:brov|edmdc)_\w+)\s*\(", txt, flags=re.M)


def test_every_declared_symbol_is_exported_and_bound(lib):
    from bluerov2_dynamics_amd import _build, _lib
    names = header_symbols()
    assert len(names) >= 30 and len(set(names)) == len(names)
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _build.LIB], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported, set(names) - exported
    # nothing but the C ABI leaks out of the shared object (no torch / C++ types in the boundary)
    assert all(s.startswith(("brov_", "edmdc_")) for s in exported if not s.startswith("_")), exported
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.brov_abi_version() == 1
    assert [lib.brov_model_nx(m) for m in (0, 1, 2)] == [12, 12, 13]
    assert [lib.brov_model_nu(m) for m in (0, 1, 2)] == [8, 6, 6]
    assert lib.brov_model_nx(7) < 0


def test_struct_layout_matches_header():
    from bluerov2_dynamics_amd import _lib
    # 4+3+3 scalars, 3x6, 3 current, 2x24 geometry, 5 poly, 9+3+3 lag = 99 doubles
    assert ctypes.sizeof(_lib.BrovParams) == 99 * 8
    p = _lib.default_params()
    assert (p.m, p.g, p.volume, p.zb) == (13.5, 9.82, 0.0134, -0.01)
    assert list(p.thrust_poly) == [8.9, 176.0, -404.1, 389.9, -140.3]
    assert list(p.lag_Cc) == [0.0, 5.992, 3.317]


def test_host_discretisation_matches_scipy_fixture(lib):
    from bluerov2_dynamics_amd import _lib
    g = load_golden("fossen_constants.npz")
    for dt in g["dts"]:
        Ad, Bd = _lib.discretise_lag(float(dt))
        assert rel_err(Ad, g[f"Ad_{dt}"]) < 1e-14 and rel_err(Bd, g[f"Bd_{dt}"]) < 1e-14
    with pytest.raises(_lib.BrovError):
        _lib.discretise_lag(-1.0)
    with pytest.raises(_lib.BrovError):
        _lib.discretise_lag(float("nan"))


def test_derived_constants_match_reference_fixture(lib):
    from bluerov2_dynamics_amd import _lib
    g = load_golden("fossen_constants.npz")
    Minv, T = _lib.derived()
    assert rel_err(Minv, np.diag(g["Minv"])) < 1e-15
    assert rel_err(T, g["alloc"]) < 1e-15
    p = _lib.default_params()
    assert rel_err(np.array([list(r) for r in p.thr_r]), g["thr_r"]) < 1e-15
    assert rel_err(np.array([list(r) for r in p.thr_dir]), g["thr_dir"]) < 1e-15


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    from bluerov2_dynamics_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.BrovError, match="no CPU fallback"):
        _lib.Context(0)
    from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2
    with pytest.raises(_lib.BrovError):
        BlueROV2()
    from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
    m = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=4)
    with pytest.raises(_lib.BrovError):
        m.fit(np.zeros((10, 12)), np.zeros((10, 8)), centers=np.zeros((4, 12)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "bluerov2_dynamics_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "/root/reference" not in src, f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/brov2.h is the drop-in boundary: it must compile as C99 and C++11 with no HIP or torch headers, and a
    plain C program must link against libbrov2.so and reach the GPU-free entry points."""
    import shutil
    import subprocess
    from conftest import REPO
    from bluerov2_dynamics_amd import _build
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    inc = os.path.join(REPO, "include")
    src = tmp_path / "t.c"
    src.write_text(
        '#include <stdio.h>\n#include "brov2.h"\n'
        "int main(void) {\n"
        "  brov_params p; double Ad[9], Bd[3];\n"
        "  brov_default_params(&p);\n"
        "  if (brov_abi_version() != BROV2_ABI_VERSION) return 2;\n"
        "  if (brov_model_nx(BROV_WRENCH_QUAT) != 13 || brov_model_nu(BROV_THRUSTER_EULER) != 8) return 3;\n"
        "  if (brov_discretise_lag(&p, 0.02, Ad, Bd) != BROV_OK) return 4;\n"
        '  printf("%.17g %.17g %.6f\\n", Ad[0], Bd[0], p.m);\n'
        "  return 0;\n}\n")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)]):
        subprocess.check_call(cmd)
    exe = tmp_path / "t"
    libdir = os.path.dirname(_build.LIB)
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-l:libbrov2.so", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).split()
    g = load_golden("fossen_constants.npz")
    assert abs(float(out[0]) - g["Ad_0.02"][0, 0]) < 1e-15 and abs(float(out[1]) - g["Bd_0.02"][0]) < 1e-15 and float(out[2]) == 13.5
