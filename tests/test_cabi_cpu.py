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


def test_arch_gate_and_comm_entry_points(lib):
    """brov_create refuses any device that is not gfx950 (BROV_ERR_NODEVICE): the predicate it applies is exported, so the
    negative case is testable without such a device.  The RCCL entry points bind (dlopen) without a GPU."""
    assert lib.brov_arch_is_supported(b"gfx950:sramecc+:xnack-") == 1
    assert lib.brov_arch_is_supported(b"gfx950") == 1
    for other in (b"gfx942:sramecc+:xnack-", b"gfx90a", b"gfx9500", b"", b"sm_90"):
        assert lib.brov_arch_is_supported(other) == 0, other
    assert lib.brov_arch_is_supported(None) == 0
    assert lib.brov_comm_available() in (0, 1)
    h = ctypes.c_void_p()
    ident = (ctypes.c_ubyte * 128)()
    assert lib.brov_comm_init_rank(0, ctypes.addressof(ident), 0, 0, ctypes.byref(h)) == -1      # nranks < 1: BROV_ERR_ARG
    assert lib.brov_comm_init_rank(0, None, 1, 0, ctypes.byref(h)) == -1
    assert lib.brov_comm_nranks(None) == -1 and lib.edmdc_gram_allreduce_dev(None, None, 0, None, 0, None) == -1


def test_missing_rccl_is_reported_not_a_crash(lib):
    """A host without librccl: brov_comm_available() == 0 and the brov_comm_* entry points return BROV_ERR_COMM -- the
    process must survive (round-2 advice: the not-found message called dlerror() twice and built a std::string from the
    NULL the second call returns).  Fresh process, raw ctypes, BROV2_RCCL_LIBRARY (the only candidate when set) bogus."""
    from bluerov2_dynamics_amd import _build
    code = (
        "import ctypes, sys\n"
        f"lib = ctypes.CDLL({_build.LIB!r})\n"
        "assert lib.brov_comm_available() == 0\n"
        "ident = (ctypes.c_ubyte * 128)()\n"
        "assert lib.brov_comm_unique_id(ident) == -5, lib.brov_comm_unique_id(ident)\n"
        "h = ctypes.c_void_p()\n"
        "assert lib.brov_comm_init_rank(0, ident, 1, 0, ctypes.byref(h)) == -5\n"
        "assert lib.brov_abi_version() == 1\n"
        "print('survived')\n")
    env = dict(os.environ, BROV2_RCCL_LIBRARY="/nonexistent/librccl-not-here.so")
    r = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_gram_decomposition_is_host_only_and_fills_the_chip(lib):
    """edmdc_gram_decomposition (no device work): the benchmark shape packs its 1 683 wanted tile products into 73 tasks (the x
    part of Y rides in the tail tile's padding); a shape whose padding is too narrow keeps the 33rd Y tile (76 tasks); tasks x
    slabs never exceeds the 2 048 wave slots of one resident round; bad shapes are refused."""
    from bluerov2_dynamics_amd import engine
    assert engine.gram_decomposition(12, 8, 512) == (73, 28)
    assert engine.gram_decomposition(13, 8, 512) == (76, 26)
    for n, r, k in ((12, 8, 48), (12, 8, 200), (12, 8, 500), (12, 6, 512), (13, 6, 512), (9, 4, 100), (12, 8, 1024), (5, 2, 16)):
        nt, ns = engine.gram_decomposition(n, r, k)
        tiles_g = ((k + 15) // 16 * 16 + (n + r + 15) // 16 * 16) // 16
        assert nt * 24 >= tiles_g * (tiles_g + 1) // 2 + tiles_g * ((k + 15) // 16)        # at least the wanted products
        assert 1 <= ns <= 256 and (nt * ns <= 2048 or ns == 1)
    # fit()'s apply pass (round 3): W rows in 17 blocks per 192 rows with all 408 tile products wanted (34 tiles = 5 x 6 + 4: both
    # block orientations), W^T Y in 49 tasks x 41 slabs (54 x 37 with plain 4 x 6 blocks); every shape keeps one resident round
    assert engine.apply_decomposition(12, 8, 512) == dict(wrows_items_per_192_rows=17, wrows_tiles_wanted=408, wty_tasks=49, wty_slabs=41)
    for n, r, k in ((12, 8, 48), (12, 8, 64), (12, 8, 96), (12, 8, 200), (13, 6, 500), (9, 4, 100), (12, 8, 1024), (5, 2, 16)):
        dec = engine.apply_decomposition(n, r, k)
        tiles_w = ((k + 15) // 16 * 16 + (n + r + 15) // 16 * 16) // 16
        tiles_y = (k + 15) // 16 + (n + 15) // 16
        assert dec["wrows_items_per_192_rows"] * 24 >= dec["wrows_tiles_wanted"] == 12 * tiles_w
        assert dec["wrows_items_per_192_rows"] * 24 < dec["wrows_tiles_wanted"] + 4 * 24           # at most a ragged column group
        assert dec["wty_tasks"] * 24 >= tiles_w * tiles_y and (dec["wty_tasks"] * dec["wty_slabs"] <= 2048 or dec["wty_slabs"] == 1)
    # fit()'s Gram pass: G^T G alone, 595 wanted tile products in 28 tasks x 73 slabs (the full Gram: 1 683 in 73 x 28)
    assert engine.gtg_decomposition(12, 8, 512) == (28, 73)
    for n, r, k in ((12, 8, 48), (12, 8, 200), (13, 6, 500), (12, 8, 1024), (5, 2, 16)):
        nt_, ns_ = engine.gtg_decomposition(n, r, k)
        tiles_g = ((k + 15) // 16 * 16 + (n + r + 15) // 16 * 16) // 16
        assert nt_ * 24 >= tiles_g * (tiles_g + 1) // 2 and (nt_ * ns_ <= 2048 or ns_ == 1) and nt_ <= engine.gram_decomposition(n, r, k)[0]
    nt, ns = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.edmdc_apply_decomposition(0, 8, 512, ctypes.byref(nt), ctypes.byref(ns), ctypes.byref(nt), ctypes.byref(ns)) == -1
    assert lib.edmdc_gram_decomposition(0, 8, 512, ctypes.byref(nt), ctypes.byref(ns)) == -1
    assert lib.edmdc_gram_decomposition(12, 8, 0, ctypes.byref(nt), ctypes.byref(ns)) == -1


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus N` outside torch.distributed.run launches its own ranks or fails loudly; with fewer than N GPUs
    visible (none here) it must exit non-zero before touching anything."""
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the self-launch would really run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BROV2_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 2 and "refusing" in out.stderr, (out.returncode, out.stderr[-300:])
    # a world size that contradicts --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    if not torch.cuda.is_available():
        return       # (the rank path needs a GPU to get as far as the check on a CPU-only host only via set_device: skip)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env2)
    assert out.returncode == 2


def test_bench_instruction_counts_match_the_compiler_listing(tmp_path):
    """bench.py prices the rollout kernel's issue-slot utilisation with the number of fp64 VALU instructions a trajectory-wave
    executes per step; re-derive that number from the compiler's own listing of the shipped source (tools/isa_loops.py):
    body loop + thrust loop of rollout_pair_kernel, minus the blocks that run rarely (the 81-instruction full sin/cos
    refresh every 64 steps and the range-extension paths of trig_delta are in the static listing)."""
    import ast
    import importlib.util
    import sys
    from bluerov2_dynamics_amd import _build
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    asm = tmp_path / "rollout.s"
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBROV2_BUILDING=1", "--offload-device-only", "-S",
                           "-o", str(asm), os.path.join(_build.CSRC, "rollout.hip")], stderr=subprocess.DEVNULL)
    for integ, pat in (("rk4", "_ZN4brov19rollout_pair_kernelILi1ELi2ELi0ELb0ELb0E"), ("euler", "_ZN4brov19rollout_pair_kernelILi0ELi2ELi0ELb0ELb0E")):
        out = subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "isa_loops.py"), str(asm), pat], text=True)
        loops = [ast.literal_eval(l.split("):", 1)[1].strip()) for l in out.splitlines() if l.startswith("loop lines")]
        assert len(loops) >= 2, out
        static = loops[0]["f64"] + loops[1]["f64"]
        rare = 81 + (45 + 20 if integ == "rk4" else 15 + 5)  # refresh block + cold range-extension code + the cos(theta) clamp behind its wave-level test, give or take
        assert static - rare - 25 <= bench.ROLLOUT_EXEC_FP64_INSTR[integ] <= static - rare + 25, (integ, static, out)
        # the two waves of a SIMD keep nothing in scratch and no accumulator-file spills
        assert all(l.get("acc", 0) == 0 and l.get("lane", 0) <= 4 for l in loops[:2]), out


def test_hand_scheduled_mfma_loops_have_no_copies_or_spills(tmp_path):
    """gram_kernel and wrows_kernel issue their operand loads as inline-assembly global_load_dwordx2 and wait with explicit
    s_waitcnt vmcnt(N): the compiler does not know the destination registers are written asynchronously (round-2 advice).
    Check on the compiler's own listing of the shipped source that no copy, spill or accumulator-file move touches an
    operand or accumulator register inside those loops, and that nothing reads a loaded register before the wait that
    covers it (tools/isa_async_loads.py; a deliberately broken listing must be flagged)."""
    import sys
    from bluerov2_dynamics_amd import _build
    asm = tmp_path / "edmdc.s"
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBROV2_BUILDING=1", "--offload-device-only", "-S",
                           "-o", str(asm), os.path.join(_build.CSRC, "edmdc.hip")], stderr=subprocess.DEVNULL)
    tool = os.path.join(REPO, "tools", "isa_async_loads.py")
    r = subprocess.run([sys.executable, tool, str(asm), "gram_kernel", "wrows_kernel"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("1 hand-scheduled loop(s) checked, 0 problem(s)") == 3, r.stdout + r.stderr
    # the checker itself: a copy of a load destination placed right behind the load must be reported
    lines = asm.read_text().split("\n")
    k = next(i for i, l in enumerate(lines) if l.startswith("_ZN4brov12wrows_kernel"))
    j = next(i for i in range(k, len(lines)) if "global_load_dwordx2" in lines[i] and "Inner Loop" in "".join(lines[i - 8:i]))
    dst = re.search(r"global_load_dwordx2 (v\[\d+:\d+\])", lines[j]).group(1)
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join(lines[:j + 1] + [f"\tv_mov_b64_e32 v[2:3], {dst}"] + lines[j + 1:]))
    r = subprocess.run([sys.executable, tool, str(bad), "wrows_kernel"], capture_output=True, text=True)
    assert r.returncode == 1 and "PROBLEM" in r.stdout, r.stdout


def test_lloyd_lds_kernel_listing(tmp_path):
    """kmeans_assign_lds_kernel (csrc/kmeans.hip) on the compiler's own listing: 4 waves per SIMD without scratch at n = 12 (the
    benchmark's instantiation; a few spilled dwords are tolerated in the generic one), every DPP chain seeded behind its
    `s_nop 1` (a DPP read needs two wait states after a VALU write of its source and the compiler does not look into the asm),
    and the evaluation loops wait for their LDS reads with an exact count -- never lgkmcnt(0) between two pairs of candidates."""
    from bluerov2_dynamics_amd import _build
    asm = tmp_path / "kmeans.s"
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBROV2_BUILDING=1", "--offload-device-only", "-S",
                           "-o", str(asm), os.path.join(_build.CSRC, "kmeans.hip")], stderr=subprocess.DEVNULL)
    lines = asm.read_text().split("\n")
    # (round 4: the single-reference filter added state; at n = 12 eleven loop-invariant dwords are parked in scratch at kernel
    # entry and re-read only inside the mask-form fallback -- never in an evaluation loop, which the walk below asserts)
    # (... and the list form of the distance bounds, LIST = true: the same budget in round 4; round 5 added the state of the dynamic
    # assignment -- tickets, their counters, the two-region list -- and the runner-up's score: 192 / 196 bytes, still none of it inside
    # an evaluation loop)
    for ns, lst, max_scratch in ((12, 0, 96), (13, 0, 128), (0, 0, 160), (12, 1, 208), (13, 1, 208)):
        k = next(i for i, l in enumerate(lines) if l.startswith(f"_ZN4brov24kmeans_assign_lds_kernelILi{ns}ELb{lst}E"))
        e = next(i for i in range(k, len(lines)) if lines[i].startswith(".Lfunc_end"))
        info = {m.group(1): int(m.group(2)) for m in (re.match(r"; (\w+): (\d+)", l) for l in lines[e:e + 40]) if m}
        assert info["Occupancy"] == 4 and info["ScratchSize"] <= max_scratch and info["NumVgprs"] <= 128, (ns, info)
        body = [l.strip() for l in lines[k:e]]
        movs = [i for i, l in enumerate(body) if l.startswith("v_mov_b64_dpp")]
        assert len(movs) >= 8 + 4                      # eight label groups + two pairs in each of the two evaluation loops
        for i in movs:
            prev = next(body[j] for j in range(i - 1, 0, -1) if body[j] and not body[j].startswith(";"))
            assert prev.startswith("s_nop 1") or prev.startswith("v_mov_b64_dpp"), (ns, body[i - 3:i + 1])
        # evaluation loops: the innermost loops that hold DPP FMAs
        heads = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
        n_eval = n_whole = 0
        for h in heads:
            end = next(i for i in range(h, len(body)) if body[i].startswith("s_cbranch"))
            loop = body[h:end]
            if sum(l.startswith("v_fmac_f64_dpp") for l in loop) >= 4 * (12 if ns else 15):
                n_eval += 1
                waits = [l for l in loop if l.startswith("s_waitcnt") and "lgkmcnt" in l]
                assert waits, (ns, lst)
                n_whole += any("lgkmcnt(0)" in w for w in waits)
                assert not any(l.startswith("scratch_") for l in loop), (ns, [l for l in loop if l.startswith("scratch_")])
        # (n = 13, plain form: the compiler merges the waits of ONE of the five inlined copies of the loop -- the fp64 walk behind an
        # uncertified packed-fp32 screening, 0.3 % of the waves -- since the kernel grew its bounds output; the benchmark's n = 12 has none)
        assert n_whole <= (1 if (ns, lst) == (13, 0) else 0), (ns, lst, n_whole)
        assert n_eval >= 2, (ns, n_eval)                 # list form (with and without the tie flags, inlined where used) + the full scan
        # the fused integer reduction of the member sums: DPP additions behind one s_nop (n = 12 / 13)
        if ns in (12, 13) and not lst:               # (the list form sends only the CHANGES of a moved sample: plain atomics)
            adds = [i for i, l in enumerate(body) if l.startswith("v_add_co_u32_dpp")]
            assert len(adds) >= 6 * ns                   # six steps over the whole wave
            first = adds[0]
            prev = next(body[j] for j in range(first - 1, 0, -1) if body[j] and not body[j].startswith(";"))
            assert prev.startswith("s_nop 1"), body[first - 3:first + 1]


def test_asynchronous_scalar_loads_are_left_alone_until_their_wait(tmp_path):
    """csrc/kmeans.hip requests a pair record with three s_load instructions from one inline-asm statement (pk_issue) and waits for it in
    another (pk_wait); in between the compiler believes the destination SGPRs hold live values.  Round 4 faulted on a destination that
    shared registers with the address pair (fixed with early-clobber outputs); nothing in the source guards the rest of that assumption
    against a compiler upgrade.  tools/isa_sload_window.py walks the compiler's own listing of the shipped source: no destination range
    overlaps its address pair or another destination, and between the requests and the s_waitcnt lgkmcnt(0) that covers them no
    instruction reads, copies, spills or overwrites a destination register and no control flow leaves the straight line.  Both kernels that
    use the pattern (kmeans_assign_lds_kernel, kmeans_assign_pk_kernel); two deliberately broken listings must be flagged.
    The other multi-instruction asm blocks of the file were audited by hand this round (csrc/kmeans.hip, comment above score_bcast)."""
    import sys
    from bluerov2_dynamics_amd import _build
    asm = tmp_path / "kmeans.s"
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBROV2_BUILDING=1", "--offload-device-only", "-S",
                           "-o", str(asm), os.path.join(_build.CSRC, "kmeans.hip")], stderr=subprocess.DEVNULL)
    tool = os.path.join(REPO, "tools", "isa_sload_window.py")
    r = subprocess.run([sys.executable, tool, str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) asynchronous pair-record request\(s\) in (\d+) kernel\(s\) checked, 0 problem", r.stdout)
    assert m and int(m.group(1)) >= 12 and int(m.group(2)) >= 4, r.stdout          # LDS / DPP kernel (n = 12, 13, list and plain) and the pk kernel
    for sub in ("kmeans_assign_lds_kernel", "kmeans_assign_pk_kernel"):
        rr = subprocess.run([sys.executable, tool, str(asm), sub], capture_output=True, text=True)
        assert rr.returncode == 0 and " 0 problem" in rr.stdout, (sub, rr.stdout)
    lines = asm.read_text().split("\n")
    pat = re.compile(r"\ts_load_dwordx16 s\[(\d+):(\d+)\], s\[(\d+):(\d+)\], 0x0$")
    i = next(i for i, l in enumerate(lines) if pat.match(l) and "s_load_dwordx8" in lines[i + 1] and "s_load_dwordx4" in lines[i + 2])
    mm = pat.match(lines[i])
    bad = tmp_path / "bad1.s"                          # a copy of a register whose load is still in flight
    bad.write_text("\n".join(lines[:i + 3] + [f"\ts_mov_b32 s99, s{mm.group(1)}"] + lines[i + 3:]))
    r1 = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r1.returncode == 1 and "while the load is in flight" in r1.stdout, r1.stdout
    a0 = int(mm.group(3))
    l2 = list(lines)                                   # the third destination on top of the address pair (the fault of round 4)
    lo = a0 - a0 % 4                                   # (the aligned range of four that holds the address pair: the pair may be s[0:1])
    l2[i + 2] = re.sub(r"s\[\d+:\d+\], s\[", f"s[{lo}:{lo + 3}], s[", l2[i + 2], count=1)
    bad2 = tmp_path / "bad2.s"
    bad2.write_text("\n".join(l2))
    r2 = subprocess.run([sys.executable, tool, str(bad2)], capture_output=True, text=True)
    assert r2.returncode == 1 and "overlaps the address pair" in r2.stdout, r2.stdout


@pytest.mark.parametrize("mode", ["auto", "0", "1"])
def test_loading_the_library_and_the_dropin_modules_does_not_import_torch(lib, mode):
    """north_star: host code = Python over a thin ctypes C ABI, torch ONLY on the bluerov_torch / PINc path.  The reference's Koopman
    module depends on numpy + scikit-learn alone (Koopman/koopmanEDMDc.py:17,26-30) and two of its scripts never import torch
    (training/train_tank_brov2_koopmanEDMDc.py:12-17).  A fresh process that imports the drop-in modules and binds the library must not
    have torch in sys.modules (BROV2_TORCH=auto / 0); BROV2_TORCH=1 restores the import-first behaviour of rounds 1-5; and with "auto" a
    later `import torch` must still find ONE HIP runtime: the libamdhip64.so mapped into the process is torch's own file."""
    import sys
    code = r'''
import sys, json
from bluerov2_dynamics_amd import _lib
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
from bluerov2_dynamics_amd.fossen import BlueROV2, BlueROV2_thrust, BlueROV2_wrench
_lib.load_library()
m = KoopmanEDMDc(state_dim=12, input_dim=8)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
print(json.dumps({"torch": "torch" in sys.modules, "hip_runtime": _lib.hip_runtime, "amdhip": sorted(set(maps))}))
'''
    env = dict(os.environ, BROV2_TORCH=mode, PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(d["amdhip"]) == 1, d                        # one HIP runtime in the process
    if mode == "1":
        assert d["torch"] and "imported by load_library" in d["hip_runtime"]
        return
    assert d["torch"] is False, d
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except ImportError:
        spec = None
    torch_hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec else None
    if mode == "auto" and torch_hip and os.path.exists(torch_hip):
        assert "preloaded" in d["hip_runtime"] and os.path.samefile(d["amdhip"][0], torch_hip), d
    else:
        assert d["hip_runtime"].startswith("system") and (torch_hip is None or not os.path.samefile(d["amdhip"][0], torch_hip)), d


def test_warm_up_is_opt_in_and_never_raises_in_the_background(lib):
    """bluerov2_dynamics_amd.warm_up(): context creation in a background thread (overlaps the 0.15-0.2 s of HIP start-up with the script's
    own start-up).  Nothing happens at import time; on a box without a GPU the thread ends quietly and the first real use raises as
    before; with a GPU the context it made is the process-wide one."""
    import bluerov2_dynamics_amd as pkg
    from bluerov2_dynamics_amd import _lib
    have_gpu = True
    try:
        _lib.Context(0).close()
    except _lib.BrovError:
        have_gpu = False
    before = dict(_lib._default)
    t = pkg.warm_up()
    t.join(60)
    assert not t.is_alive()
    if have_gpu:
        assert 0 in _lib._default and pkg.warm_up(block=True) is _lib._default[0]
    else:
        assert _lib._default == before
        with pytest.raises(_lib.BrovError):
            _lib.default_context(0)
