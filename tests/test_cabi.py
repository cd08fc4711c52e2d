"""The C-ABI library loads, exports every symbol include/psa_rk4.h declares, and validates arguments before
touching a device (so these checks run without a GPU; no compute call is made here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import psa_amd._native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psa_rk4.h")


def _declared_symbols():
    src = open(HEADER, encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psa_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    L = nat.lib()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(L, name), f"{name} declared in psa_rk4.h but not exported"
    assert sorted(nat.EXPORTED_SYMBOLS) == declared          # the ctypes table covers the whole header
    assert nat.version().startswith("psa-hip") and "gfx950" in nat.version()
    assert isinstance(nat.device_count(), int)


def test_flag_values_match_the_header():
    src = open(HEADER, encoding="utf-8").read()
    for name, val in (("PSA_BCAST_GAMMA", nat.BCAST_GAMMA), ("PSA_BCAST_ALPHA", nat.BCAST_ALPHA),
                      ("PSA_BCAST_A0", nat.BCAST_A0), ("PSA_OPT_CHECK_NAN", nat.OPT_CHECK_NAN),
                      ("PSA_OPT_EXACT_STEP", nat.OPT_EXACT_STEP), ("PSA_OPT_LDS_STAGING", nat.OPT_LDS_STAGING),
                      ("PSA_OPT_BLOCK64", nat.OPT_BLOCK64), ("PSA_OPT_F32_SCALAR", nat.OPT_F32_SCALAR),
                      ("PSA_OPT_F32_PACKED", nat.OPT_F32_PACKED), ("PSA_OPT_LOSSLESS", nat.OPT_LOSSLESS),
                      ("PSA_OPT_SPLIT_POINT", nat.OPT_SPLIT_POINT), ("PSA_OPT_ONE_LANE", nat.OPT_ONE_LANE),
                      ("PSA_OPT_TRAJ_LD", nat.OPT_TRAJ_LD), ("PSA_OPT_QUAD_POINT", nat.OPT_QUAD_POINT)):
        m = re.search(rf"#define\s+{name}\s+\(1u\s*<<\s*(\d+)\)", src)
        assert m and (1 << int(m.group(1))) == val, name


def test_n_saved_rule():
    L = nat.lib()
    assert L.psa_n_saved(1005, 10) == 101 and L.psa_n_saved(10, 2) == 6 and L.psa_n_saved(7, 10) == 1
    assert L.psa_n_saved(10, 0) == -1
    # psa_traj_ld: padded only where the wave regions would be a multiple of 2 MiB apart
    assert L.psa_traj_ld(131072, 8) == 131072 + 272 and L.psa_traj_ld(131072, 4) == 131072 and L.psa_traj_ld(262144, 4) == 262144 + 544
    assert L.psa_traj_ld(65536, 8) == 65536 and L.psa_traj_ld(0, 8) == 0 and L.psa_traj_ld(-1, 8) == -1 and L.psa_traj_ld(8, 2) == -1


def _call_dev(**over):
    """psa_rk4_sweep_f64_dev with dummy non-NULL pointers; must fail in validation, before any launch."""
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)
    a = dict(n_waves=4, n_points=8, n_steps=10, z_max=1.0, save_every=1, dbeta2=None)
    a.update(over)
    return nat.lib().psa_rk4_sweep_f64_dev(None, a["n_waves"], a["n_points"], a["n_steps"], a["z_max"], a["save_every"],
                                           p, a["dbeta2"], p, p, p, 0, p, p, p, p, None)


@pytest.mark.parametrize("over,code", [(dict(n_waves=5), -1), (dict(n_points=-1), -2), (dict(n_steps=0), -3),
                                       (dict(n_steps=2**31), -3), (dict(z_max=0.0), -4), (dict(z_max=float("nan")), -4),
                                       (dict(save_every=0), -5), (dict(n_waves=6), -8)])
def test_argument_errors_are_negative_codes_with_a_message(over, code):
    assert _call_dev(**over) == code
    assert len(nat.lib().psa_last_error()) > 0


def test_null_pointer_and_dbeta2_rules():
    L = nat.lib()
    assert L.psa_rk4_sweep_f64_dev(None, 4, 8, 10, 1.0, 1, None, None, None, None, None, 0, None, None, None, None,
                                   None) == -6
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)
    assert L.psa_rk4_sweep_f64_dev(None, 4, 8, 10, 1.0, 1, p, p, p, p, p, 0, p, p, p, p, None) == -8   # dbeta2 with 4 waves
    # n_points == 0 is a valid no-op (empty sweep) on both faces, with or without a device
    assert L.psa_rk4_sweep_f64_dev(None, 4, 0, 10, 1.0, 1, None, None, None, None, None, 0, None, None, None, None,
                                   None) == 0
    assert L.psa_rk4_sweep_f64(0, 4, 0, 10, 1.0, 1, None, None, None, None, None, 0, None, None, None, None, None,
                               None) == 0
    assert L.psa_yaman_rhs_f64(0, 0, None, None, None, None, None, None, None, None, None) == 0
    assert L.psa_gain_summary_workspace_bytes(1000) > 0


def test_python_wrapper_shape_checks():
    a0 = np.ones(4, complex)
    with pytest.raises(ValueError):
        nat.sweep_host(np.zeros((2, 2)), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=a0)
    with pytest.raises(ValueError):
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=[1.0, 2.0], alpha=0.0, a0=a0)
    with pytest.raises(ValueError):
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=np.ones(5, complex))
    with pytest.raises(ValueError):
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=np.ones(6, complex))
    with pytest.raises(ValueError):
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=a0, dtype=np.float16)


@pytest.mark.skipif(nat.device_count() > 0, reason="CPU box only")
def test_host_entry_points_report_missing_device_not_a_fallback():
    with pytest.raises(nat.PsaNativeError) as e:
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=np.ones(4, complex))
    assert e.value.code == -7 and "no CPU fallback" in str(e.value)
    with pytest.raises(nat.PsaNativeError):
        nat.gain_summary_host(np.ones(4), None, 1.0)


def test_bench_and_entry_scripts_import_on_a_cpu_box():
    """bench.py / __graft_entry__.py must at least parse and expose their contract without a GPU."""
    import ast
    import subprocess
    import sys
    for name in ("bench.py", "__graft_entry__.py"):
        ast.parse(open(os.path.join(ROOT, name), encoding="utf-8").read())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.build) and callable(mod.smoke)


def test_missing_shared_library_is_a_loud_error(monkeypatch, tmp_path):
    """No libpsa_hip.so => NativeUnavailableError naming the build command -- never a silent CPU path."""
    monkeypatch.setattr(nat, "_LIB", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "libpsa_hip.so"))
    with pytest.raises(nat.NativeUnavailableError, match="no CPU fallback"):
        nat.sweep_host(np.zeros(3), n_steps=1, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=np.ones(4, complex))
    from psa_amd import config, simulation
    from psa_amd.phase_matching import PhaseMatchingConfig
    with pytest.raises(nat.NativeUnavailableError):
        simulation.run_single_simulation(config.custom_simulation_config(z_max=1.0, dz=0.1), gamma=1.0, alpha=0.0,
                                         omega=[1.0] * 4, p_in=[1, 1, 0, 0],
                                         phase_matching_cfg=PhaseMatchingConfig(method="provided", provided_delta_beta=0.0))


def test_header_is_valid_c99_and_a_c_program_links_and_gets_the_documented_codes(tmp_path):
    """The boundary is a C ABI: compile include/psa_rk4.h as strict C99, build tests/c/abi_client.c with gcc against
    libpsa_hip.so and run it (argument validation only -- no GPU needed)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    inc = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(inc, "psa_rk4.h")], check=True)
    libdir = os.path.dirname(nat.LIB_PATH)
    exe = str(tmp_path / "abi_client")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", inc, os.path.join(ROOT, "tests", "c", "abi_client.c"),
                    "-o", exe, "-L", libdir, "-lpsa_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "abi_client ok" in out.stdout


def test_dbeta_producer_validates_its_model_before_any_launch():
    """psa_dbeta_grid_*_dev / psa_dbeta_pairs_*_dev: bad method / orders / beta count / block -> negative codes, no launch."""
    L = nat.lib()
    buf = np.zeros(16)
    p = buf.ctypes.data_as(C.c_void_p)
    beta = np.zeros(9)
    beta[2] = -1e-28
    pb = beta.ctypes.data_as(C.c_void_p)
    ok_orders = np.array([2, 4], np.int32)
    po = ok_orders.ctypes.data_as(C.c_void_p)

    def grid(method=0, orders=po, n_orders=2, max_order=4, n_beta=9, n2=2, n3=4, first=0, n=8, out=p):
        return L.psa_dbeta_grid_f64_dev(None, method, orders, n_orders, max_order, pb, n_beta, 1.2e15, 1.88e9, 0.0, 1e-12,
                                        1550e-9, p, n2, p, n3, first, n, out, None)

    assert grid(method=7) == -10
    assert grid(n_orders=0) == -10 and grid(n_orders=5) == -10 and grid(orders=None) == -10
    odd = np.array([2, 3], np.int32)
    assert grid(orders=odd.ctypes.data_as(C.c_void_p)) == -10
    big = np.array([2, 10], np.int32)
    assert grid(orders=big.ctypes.data_as(C.c_void_p)) == -10
    assert grid(n_beta=0) == -10 and grid(n_beta=10) == -10
    assert grid(method=1, max_order=9) == -10
    assert grid(first=4, n=8) == -2           # block leaves the 2 x 4 grid
    assert grid(n2=0) == -2 and grid(n=-1) == -2
    assert grid(out=None) == -6
    assert grid(n=0) == 0                       # empty block: a valid no-op, no device needed
    assert len(L.psa_last_error()) > 0
    assert L.psa_dbeta_pairs_f64_dev(None, po, 2, pb, 9, 1e12, p, 2, p, 2, 0, 5, p, p) == -2
    assert L.psa_dbeta_pairs_f64_dev(None, po, 2, pb, 9, 1e12, p, 2, p, 2, 0, 0, None, None) == 0
    assert L.psa_gain_summary_f32_dev(None, -1, p, None, 1.0, 1, p, p, p, p, p) == -2
    assert L.psa_gain_summary_f32_dev(None, 4, None, None, 1.0, 1, p, p, p, p, p) == -6


def test_n_points_beyond_the_launch_grid_is_too_large():
    """PSA_MAX_POINTS = 2^31 - 256: what a launch really takes (2^32 - 1 threads in x; the two-lane layout uses two per
    point) -- checked before any allocation or launch, so an oversized sweep gets the documented code, not a raw
    hipErrorInvalidConfiguration."""
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)
    L = nat.lib()
    src = open(HEADER, encoding="utf-8").read()
    assert int(re.search(r"#define\s+PSA_MAX_POINTS\s+(\d+)LL", src).group(1)) == nat.MAX_POINTS == 2**31 - 256
    assert L.psa_rk4_sweep_f64_dev(None, 4, nat.MAX_POINTS + 1, 10, 1.0, 1, p, None, p, p, p, 0, p, p, p, p, None) == -9
    assert L.psa_rk4_sweep_f32_dev(None, 4, 2**32, 10, 1.0, 1, p, None, p, p, p, 0, p, p, p, p, None) == -9
    assert L.psa_rk4_sweep_f64(0, 4, nat.MAX_POINTS + 1, 10, 1.0, 1, p, None, p, p, p, 0, p, p, p, p, None, None) == -9


def test_trajectory_launch_limits_and_contradictory_flags():
    """Trajectory rows are addressed with a 32-bit lane offset kept below 2^31 (float64: < 2^27 points, float32: < 2^28; two
    lanes per point: n_waves * N * 16 B < 2^32); SPLIT_POINT + ONE_LANE and F32_SCALAR + F32_PACKED are rejected."""
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)
    L = nat.lib()

    def f64(n, flags=0, nw=4, traj=p):
        return L.psa_rk4_sweep_f64_dev(None, nw, n, 10, 1.0, 1, p, p if nw == 6 else None, p, p, p, flags, p, p, p, p, traj)
    assert f64(2**27) == -9 and b"trajectory" in L.psa_last_error()
    assert L.psa_rk4_sweep_f32_dev(None, 4, 2**28, 10, 1.0, 1, p, None, p, p, p, 0, p, p, p, p, p) == -9
    assert f64(2**26, nat.OPT_SPLIT_POINT) == -9 and f64(2**26 // 6 * 4 + 8, nat.OPT_SPLIT_POINT, nw=6) == -9
    assert f64(8, nat.OPT_SPLIT_POINT | nat.OPT_ONE_LANE, traj=None) == -11
    assert f64(8, nat.OPT_QUAD_POINT | nat.OPT_ONE_LANE, traj=None) == -11 and f64(8, nat.OPT_QUAD_POINT, nw=6, traj=None) == -11
    assert f64(2**26, nat.OPT_QUAD_POINT) == -9
    assert L.psa_rk4_sweep_f32_dev(None, 4, 8, 10, 1.0, 1, p, None, p, p, p, nat.OPT_F32_SCALAR | nat.OPT_F32_PACKED,
                                   p, p, p, p, None) == -11
    assert b"exclude" in L.psa_last_error()
    assert L.psa_release_cache() >= 0


def test_dbeta_model_description_of_the_python_carriers():
    from psa_amd import dispersion, phase_matching
    d = dispersion.DispersionParams(omega_ref=1.2e15, beta2=-2e-28, beta3=4e-41, beta4=-3e-55, extra={6: 1e-84})
    m = nat.dbeta_model(d, phase_matching.PhaseMatchingConfig(even_orders=(2, 4, 6)))
    assert m["method"] == nat.DBETA_SYMMETRIC_EVEN and list(m["orders"]) == [2, 4, 6] and m["beta"][6] == 1e-84
    m = nat.dbeta_model(d, phase_matching.PhaseMatchingConfig(method="general_taylor", max_order=6))
    assert m["method"] == nat.DBETA_GENERAL_TAYLOR and m["max_order"] == 6
    with pytest.raises(ValueError):
        nat.dbeta_model(d, phase_matching.PhaseMatchingConfig(method="provided", provided_delta_beta=0.1))
    with pytest.raises(ValueError):
        nat.dbeta_model(dispersion.DispersionParams(omega_ref=1.2e15, extra={10: 1e-140}), phase_matching.PhaseMatchingConfig())
    with pytest.raises(ValueError):
        nat.dbeta_model(None, phase_matching.PhaseMatchingConfig())


def test_hot_loops_of_the_built_kernels_start_on_8_byte_boundaries():
    """The z-loops are streams of 8-byte encodings; a loop body 4 bytes off an 8-byte boundary was measured 15 % slower
    (profiles/r03_loop_alignment.log).  The sweep TUs are compiled with -mllvm -align-all-blocks=3: check the objects."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "psa-simulation-ode-rk-mvp-dispersion_amd", "csrc")
    for obj in ("psa_rk4_f64.o", "psa_rk4_f32.o"):
        path = os.path.join(csrc, obj)
        if not os.path.exists(path):
            pytest.skip("kernel objects not present (library built elsewhere)")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "loop_alignment.py"), path], capture_output=True,
                             text=True, check=True).stdout.strip().splitlines()[-1]
        m = re.search(r"(\d+) hot loops, (\d+) not 8-byte aligned", out)
        assert m and int(m.group(1)) > 100 and int(m.group(2)) == 0, out


def test_eight_byte_encodings_of_the_hot_loops_sit_on_8_byte_boundaries():
    """With one wave per SIMD an 8-byte VALU instruction 4 bytes off its boundary issues in 5 cycles instead of 4
    (tools/issue_probe.hip, profiles/r03_issue_probe.log); every 4-byte encoding the compiler emits flips the alignment
    of what follows it (26 % of the 8-byte instructions of the plain build's hot loops are off).  csrc/align_encodings.py
    re-encodes one 4-byte instruction per odd run: check the built objects (a plain fallback build fails here)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "psa-simulation-ode-rk-mvp-dispersion_amd", "csrc")
    for obj in ("psa_rk4_f64.o", "psa_rk4_f32.o"):
        path = os.path.join(csrc, obj)
        if not os.path.exists(path):
            pytest.skip("kernel objects not present (library built elsewhere)")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "encoding_alignment.py"), path, "--summary"],
                             capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
        m = re.search(r"(\d+) hot loops, (\d+) 8-byte VALU instructions in them, (\d+) off by 4", out)
        assert m and int(m.group(1)) > 100 and int(m.group(3)) <= 0.02 * int(m.group(2)), out
