"""CPU suite: the C-ABI libraries load, export every declared symbol, and the product path refuses to run
without a GPU (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, src)))


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    g.build()
    from co_fusion_amd import lib as cflib
    return cflib


def test_c_abi_exports_every_declared_symbol(libs):
    names = _declared("cofusion_hip.h", "cf_")
    assert len(names) > 50
    out = subprocess.check_output(["nm", "-D", "--defined-only", libs.LIB_PATH]).decode()
    exported = set(re.findall(r" T (\w+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in include/cofusion_hip.h but not exported: {missing}"
    assert set(libs.SYMBOLS) <= exported


def test_facade_exports_every_declared_symbol(libs):
    names = _declared("cofusion.h", "cofusion_")
    out = subprocess.check_output(["nm", "-D", "--defined-only", libs.HOST_LIB_PATH]).decode()
    exported = set(re.findall(r" T (\w+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in include/cofusion.h but not exported: {missing}"
    assert set(libs.HOST_SYMBOLS) <= exported


def test_pod_layouts_match_the_reference_abi(libs):
    """DataTerm 16 B (types.cuh:75-81), CameraModel 16 B, surfel 48 B (Vertex.cpp:43)."""
    from co_fusion_amd import api
    assert api.DATATERM.itemsize == 16
    assert C.sizeof(api.Cam) == 16


def test_product_path_fails_loudly_without_gpu(libs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from co_fusion_amd import api, facade
    with pytest.raises(api.CofusionError):
        api.Context(64, 48, 50, 50, 32, 24)
    with pytest.raises(facade.CoFusionError):
        facade.CoFusion(64, 48, 50, 50, 32, 24)
    # and the raw C-ABI reports an error instead of silently computing on the host
    lib = libs.load()
    cfg = api.Config(64, 48, 50, 50, 32, 24, 0, 1, 1024)
    h = C.c_void_p()
    assert lib.cf_create(C.byref(cfg), C.byref(h)) != 0
    # ... so do the facade's C entry points (one instance, a lock-step group), with a message
    with pytest.raises(facade.CoFusionError):
        facade.CoFusionGroup(2, 64, 48, 50, 50, 32, 24)
    host = libs.load_host()
    fcfg = facade._make_config(host, 64, 48, 50.0, 50.0, 32.0, 24.0, 0, {})
    g = C.c_void_p()
    host.cofusion_last_error.restype = C.c_char_p
    assert host.cofusion_group_create(C.byref(fcfg), 2, C.byref(g)) != 0 and not g.value
    assert host.cofusion_last_error(), "an error without a message"
    assert host.cofusion_group_create(C.byref(fcfg), 0, C.byref(g)) != 0, "a group of zero sequences"


def test_no_product_code_touches_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "co_fusion_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                # comments may cite the oracle's spec; code must not include, import, load or call it
                bad = re.search(r'#include\s*"[^"]*orc[^"]*"|^\s*(import|from)\s+orc|liborc|\borc_[a-z0-9_]+\s*\(', txt, flags=re.M)
                assert not bad, f"{f} uses the oracle: {bad.group(0)}"


def test_sqrt_gate_bounds_terminate_and_decide_exactly(libs):
    """The radicand bounds of the two ICP gates (sqrtf(x) < T, sqrtf(x) <= T decided on x; csrc/track_reduce.hip) for thresholds a
    C-ABI caller may pass to cf_icp_step: 0, a denormal, FLT_MAX and +inf (gate disabled) must not hang and must decide exactly."""
    import numpy as np
    lib = libs.load()
    lt = getattr(lib, "_ZN2cf12sqrt_gate_ltEf"); le = getattr(lib, "_ZN2cf12sqrt_gate_leEf")
    for f in (lt, le):
        f.restype = C.c_float; f.argtypes = [C.c_float]
    f32 = np.float32
    fmax = float(np.finfo(f32).max)
    for T in (0.0, 1e-42, 1e-20, 0.1, float(f32(np.sin(f32(20.0) * f32(3.14159254) / f32(180.0)))), 1.0, 1e19, fmax, float("inf")):
        bl, be = f32(lt(T)), f32(le(T))
        Tf = f32(T)
        # probe the neighbourhood of the bounds
        for b, strict in ((bl, True), (be, False)):
            xs = [b, np.nextafter(b, f32(np.inf)), np.nextafter(b, f32(0))] if np.isfinite(b) else [f32(fmax), f32(np.inf)]
            for x in xs:
                if not (x >= 0):
                    continue
                want = (np.sqrt(f32(x)) < Tf) if strict else (np.sqrt(f32(x)) <= Tf)
                got = (f32(x) < bl) if strict else (f32(x) <= be)
                assert bool(want) == bool(got), (T, float(x), strict)


def test_library_links_rccl_and_resolves_it(libs):
    """RCCL is called from inside the C-ABI library (csrc/rccl_comm.hip): it imports the nccl* entry points, and they resolve when the
    library is loaded (no GPU needed to create a unique id is NOT assumed: only symbol resolution and the argument checks are exercised
    here).  The facade library reaches RCCL only through the C-ABI (cofusion_rccl_unique_id forwards to cf_rccl_unique_id): ONE library
    depends on RCCL."""
    for path, needed, exact in ((libs.LIB_PATH, {"ncclAllReduce", "ncclBroadcast", "ncclCommInitRank", "ncclCommDestroy", "ncclGetUniqueId"}, False),
                                (libs.HOST_LIB_PATH, set(), True)):
        out = subprocess.check_output(["nm", "-D", "--undefined-only", path]).decode()
        imported = set(re.findall(r" U (nccl\w+)", out))
        assert needed <= imported and (not exact or imported == needed), f"{os.path.basename(path)} imports {sorted(imported)}"
    lib = libs.load()          # dlopen succeeded: librccl.so.1 was found and every nccl symbol bound
    host = libs.load_host()
    assert lib.cf_rccl_init(None, None, 0, 1) != 0 and lib.cf_rccl_allreduce(None, None, 0, 0, None) != 0
    assert lib.cf_rccl_destroy(None) != 0
    assert host.cofusion_init_rccl(None, None) != 0 and host.cofusion_broadcast(None, None, 0, 0) != 0


def test_division_by_the_image_width_is_exact(tmp_path):
    """the tracking kernels divide pixel indices by the image width with a multiply-high and a shift (cf_kernels.h: make_idiv, late round
    3): the host function that makes the magic numbers, compiled from the real header, against the integer division"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "idiv_check")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", "-I", os.path.join(root, "co_fusion_amd", "csrc"),
                           os.path.join(root, "tests", "native", "idiv_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr


def test_covariance_of_the_normal_matrix_matches_the_oracle(libs):
    """cf_odom_get_covariance (RGBDOdometry::getCovariance, RGBDOdometry.cpp:479: partial-pivot LU inverse of lastA) is host arithmetic:
    against the oracle's statement of it on well-conditioned, nearly singular, pivoting and all-zero matrices (the covered-sensor case of
    the -rl branch: NaN pattern included), bit for bit."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from co_fusion_amd import api
    lib = libs.load()
    rng = np.random.default_rng(11)
    mats = []
    for n in (6, 7, 40, 5000):
        J = rng.normal(size=(n, 6)) * rng.uniform(0.1, 30.0, size=6)
        mats.append(J.T @ J)
    P = np.eye(6)[[3, 0, 5, 1, 4, 2]]
    mats += [P @ mats[2], np.zeros((6, 6)), np.diag([1.0, 2.0, 0.0, 4.0, 5.0, 6.0]), mats[0] * 1e-30]
    for A in mats:
        st = api.TrackStats()
        a = np.ascontiguousarray(A, np.float64).reshape(36)
        for i in range(36):
            st.lastA[i] = a[i]
        got = np.zeros(36, np.float64); want = np.zeros(36, np.float64)
        assert lib.cf_odom_get_covariance(C.byref(st), got.ctypes.data_as(C.c_void_p)) == 0
        orc.lib.orc_covariance(a.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        assert got.tobytes() == want.tobytes() or (np.isnan(got) == np.isnan(want)).all() and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    ok = np.linalg.inv(mats[2])
    st = api.TrackStats()
    for i in range(36):
        st.lastA[i] = mats[2].reshape(36)[i]
    got = np.zeros(36)
    lib.cf_odom_get_covariance(C.byref(st), got.ctypes.data_as(C.c_void_p))
    assert np.allclose(got.reshape(6, 6), ok, rtol=1e-9, atol=0)
    assert lib.cf_odom_get_covariance(None, got.ctypes.data_as(C.c_void_p)) != 0
