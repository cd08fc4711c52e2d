"""GPU tests of the reference-shaped Python API (``-m gpu``): the same calls a user of the reference makes,
served by the HIP kernels, compared with the outputs the reference itself produced (tests/golden)."""
import numpy as np
import pytest

import psa_amd
from psa_amd import config, dispersion, frequency_plan, integrators, parameters, scan_mismtach, simulation, yaman_model
from psa_amd.phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
from conftest import ATOL_DB, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu


def _disp(g, prefix=""):
    return dispersion.DispersionParams(omega_ref=float(g[prefix + "omega_ref"]), beta2=float(g[prefix + "beta2"]),
                                       beta3=float(g[prefix + "beta3"]), beta4=float(g[prefix + "beta4"]))


def test_g1_run_single_simulation_full_trajectory(golden):
    """main.py:22-96 scenario through run_single_simulation: 10 000 steps, 1001 rows, 45.29 dB."""
    g = golden("G1")
    om = frequency_plan.plan_from_wavelengths(*g["lam"])
    sp = frequency_plan.infer_symmetry_from_omegas(*om)
    d = dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.02, 0.02, 0,
                                              D_units="ps/nm/km", S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                              omega_ref=sp.omega_c)
    cfg = config.custom_simulation_config(z_max=1000.0, dz=0.1)
    z, A = simulation.run_single_simulation(cfg, gamma=float(g["gamma"]), alpha=float(g["alpha"]), omega=om,
                                            p_in=g["p_in"], phase_in=np.zeros(4), dispersion=d,
                                            phase_matching_cfg=PhaseMatchingConfig(), length_unit="m")
    assert A.shape == (1001, 4) and A.dtype == np.complex128
    assert np.array_equal(z, g["z"])
    assert rel_err(A, g["A"]) < RTOL_F64
    gain = 10 * np.log10(abs(A[-1, 2]) ** 2 / g["p_in"][2])
    assert abs(gain - 45.292443557977066) < ATOL_DB


def test_g4_examples_km_path_and_wave_order(golden):
    """reference tests.py:318 (example_zero_signal: A[0,2] == A[0,3] == 0 exactly) + both example trajectories."""
    g = golden("G4")
    z, A = simulation.example_zero_signal()
    assert A.shape == g["zero_A"].shape and A[0, 2] == 0 and A[0, 3] == 0 and np.all(A[:, 2:] == 0)
    np.testing.assert_allclose(z, g["zero_z"], rtol=0, atol=1e-15)
    assert rel_err(A[:, :2], g["zero_A"][:, :2]) < RTOL_F64
    z, A = simulation.custom_seeded_signal()
    np.testing.assert_allclose(z, g["seeded_z"], rtol=0, atol=1e-15)
    assert rel_err(A, g["seeded_A"]) < RTOL_F64


def test_g2_gain_and_dbeta_driver(golden):
    g = golden("G2")
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=float(g["lambda_p1"]), lambda_p2_m=float(g["lambda_p2"]), lambda_signal_m=g["lambda3"],
        gamma=float(g["gamma"]), alpha=float(g["alpha"]), p_in=g["p_in"], dispersion=_disp(g), length_unit="m",
        gain_unit="dB", phase_in=np.zeros(4), show=False, show_progress=False)
    assert np.array_equal(x, g["x"]) and np.array_equal(db, g["dbeta"])
    np.testing.assert_allclose(gain, g["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    assert gain[0] == pytest.approx(9.6432746655328694e-16, rel=1e-6)      # floor points: max is the z = 0 row


def test_g3_gain_driver_45_db(golden):
    g = golden("G3")
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    x, gain = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=float(g["lambda_p1"]), lambda_p2_m=float(g["lambda_p2"]), lambda_signal_m=g["lambda3"],
        gamma=float(g["gamma"]), alpha=float(g["alpha"]), p_in=g["p_in"], phase_in=np.zeros(4), dispersion=_disp(g),
        phase_matching_cfg=PhaseMatchingConfig(), gain_unit="db", show=False, show_progress=False)
    assert np.array_equal(x, g["x"]) and not np.isnan(gain).any()
    np.testing.assert_allclose(gain, g["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    assert int(np.argmax(gain)) == 4


def test_g11_km_units_linear_gain_input_phases(golden):
    g = golden("G11")
    for tag, unit in (("m", "m"), ("km", "km")):
        z_max, dz, se, gamma, alpha = g[tag + "_cfg"]
        dv = g["disp_" + tag]
        d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
        cfg = config.custom_simulation_config(z_max=z_max, dz=dz, save_every=int(se))
        x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
            cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1556e-9, lambda_signal_m=g["lambda3"], gamma=gamma, alpha=alpha,
            p_in=g["p_in"], phase_in=g["phase_in"], dispersion=d, length_unit=unit, return_wavelength_unit="m",
            gain_unit="linear", show=False, show_progress=False)
        assert np.array_equal(x, g[tag + "_x"]) and np.array_equal(db, g[tag + "_dbeta"])
        np.testing.assert_allclose(gain, g[tag + "_gain"], rtol=RTOL_F64)


def test_g9_check_nan_raises_with_the_reference_step_index(golden):
    g = golden("G9")
    w0 = 2 * np.pi * 299792458.0 / 1.55e-6
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=float(g["dbeta"]))
    cfg = config.custom_simulation_config(z_max=100.0, dz=0.1)
    for gam, step in zip(g["gammas"], g["first_bad_step"]):
        kw = dict(gamma=float(gam), alpha=0.0, omega=np.full(4, w0), p_in=g["p_in"], phase_matching_cfg=pm)
        if step >= 0:
            with pytest.raises(FloatingPointError, match=rf"NaN or Inf detected at step {int(step)}, z = "):
                simulation.run_single_simulation(cfg, **kw)
        else:
            z, A = simulation.run_single_simulation(cfg, **kw)
            assert np.isfinite(A).all()
    cfg_off = config.custom_simulation_config(z_max=100.0, dz=0.1, check_nan=False)
    z, A = simulation.run_single_simulation(cfg_off, gamma=200.0, alpha=0.0, omega=np.full(4, w0), p_in=g["p_in"],
                                            phase_matching_cfg=pm)
    assert A.shape[0] == int(g["nocheck_n_rows"])
    assert int(np.argmax(~np.isfinite(A).all(axis=1))) == int(g["nocheck_first_bad_row"])


def test_g9_driver_nan_masks(golden):
    g = golden("G9")
    cfg = config.custom_simulation_config(z_max=100.0, dz=0.1)
    d = _disp(g, "drv_")
    x, gain = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=g["drv_lambda3"], gamma=200.0, alpha=0.0,
        p_in=g["p_in"], dispersion=d, show=False, show_progress=False)
    assert np.isnan(gain).all()                                   # every run overflows -> NaN, no exception
    x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=g["mixed_lambda3"], gamma=0.0115, alpha=0.0,
        p_in=g["p_in"], dispersion=d, show=False, show_progress=False)
    np.testing.assert_array_equal(np.isnan(gain), [False, True, False])
    np.testing.assert_allclose(gain[[0, 2]], g["mixed_gain"][[0, 2]], rtol=RTOL_F64, atol=ATOL_DB)
    assert np.isnan(db[1]) and np.array_equal(db[[0, 2]], g["mixed_dbeta"][[0, 2]])


def _params(gamma, alpha, dbeta):
    mp = parameters.make_model_params(waves=parameters.WavesParams(omega=[1.2e15] * 4),
                                      fiber=parameters.FiberParams(length_m=100.0, gamma_W_m=gamma, alpha_1_m=alpha),
                                      grid=parameters.SimulationGrid(dz_m=0.1))
    mp.cache.set_phase_mismatch(dbeta)
    return mp


def test_integrator_operator_api_with_the_native_handle(golden, oracle):
    """integrate_interval / integrate_fixed_step / rk4_step accept ``rhs_yaman_simplified`` like the reference."""
    g = golden("G7")
    a0 = np.sqrt(g["p_in"]) * np.exp(1j * g["phase_in"])
    mp = _params(float(g["gamma"]), float(g["alpha"]), float(g["dbeta"]))
    z, A = integrators.integrate_interval(yaman_model.rhs_yaman_simplified, 100.5, 0.1, a0, mp, save_every=10)
    assert np.array_equal(z, g["n1005_se10_z"]) and rel_err(A, g["n1005_se10_A"]) < RTOL_F64
    z2, A2 = integrators.integrate_fixed_step(yaman_model.rhs_yaman_simplified, np.linspace(0, 100.5, 1006), a0, mp,
                                              save_every=10)
    assert np.array_equal(A2, A)
    # one explicit step through the callable handle (4 launches of the batched RHS kernel)
    y1 = integrators.rk4_step(yaman_model.rhs_yaman_simplified, 0.0, a0, 0.1, mp)
    zr, Ar, _ = oracle.integrate(a0, z_max=0.1, n=1, save_every=1, gamma=float(g["gamma"]), alpha=float(g["alpha"]),
                                 dbeta=float(g["dbeta"]))
    assert rel_err(y1, Ar[-1]) < 1e-13
    # a non-uniform grid cannot use the in-kernel z-loop: it goes through the callable contract, same numbers
    grid = np.concatenate([np.linspace(0, 1.0, 11), [1.25, 1.5]])
    z3, A3 = integrators.integrate_fixed_step(yaman_model.rhs_yaman_simplified, grid, a0, mp, save_every=1)
    assert A3.shape == (13, 4) and np.isfinite(A3).all()


def test_rhs_callable_matches_g5(golden):
    g = golden("G5")
    for i in range(0, 64, 7):
        mp = _params(float(g["gamma"][i]), float(g["alpha"][i]), float(g["dbeta"][i]))
        out = yaman_model.rhs_yaman_simplified(float(g["z"][i]), g["a"][i], mp)
        assert out.shape == (4,) and out.dtype == np.complex128
        assert np.max(np.abs(out - g["rhs"][i])) <= 1e-14 * np.max(np.abs(g["rhs"][i]))
    with pytest.raises(ValueError, match="shape"):
        yaman_model.rhs_yaman_simplified(0.0, np.ones(3, complex), _params(1.0, 0.0, 0.0))
    lin, kerr, fwm = yaman_model.yaman_terms(g["z"], g["a"], g["gamma"], g["alpha"], g["dbeta"])
    assert rel_err(kerr, g["kerr"]) < 1e-13


def test_direct_dbeta_scan_end_and_max_modes(golden):
    """The working counterpart of the reference's dead scan_mismatch_seeded_signal, pinned by G8."""
    g = golden("G8")
    cfg = config.custom_simulation_config(z_max=1000.0, dz=0.1)
    for mode, key in (("end", "n1e4_a1_p_end"), ("max", "n1e4_a1_p_max")):
        r = scan_mismtach.scan_dbeta_seeded_signal(cfg=cfg, delta_beta=g["dbeta257"], gamma=float(g["gamma"]),
                                                   alpha=float(g["alphas"][1]), p_in=g["p_in"], gain_mode=mode,
                                                   gain_unit="linear")
        ref = g[key] / g["p_in"][2]
        np.testing.assert_allclose(r["gain"], ref, rtol=RTOL_F64)
        assert r["best_index"] == int(np.argmax(ref)) and r["n_finite"] == 257
        assert r["best_delta_beta"] == g["dbeta257"][r["best_index"]]
    # km units: dbeta, gamma, alpha x1000 and lengths /1000 give the same gains
    cfg_km = config.custom_simulation_config(z_max=1.0, dz=1e-4)
    rk = scan_mismtach.scan_dbeta_seeded_signal(cfg=cfg_km, delta_beta=g["dbeta257"] * 1e3, gamma=11.5, alpha=0.115,
                                                p_in=g["p_in"], length_unit="km", gain_mode="max", gain_unit="linear")
    np.testing.assert_allclose(rk["gain"], g["n1e4_a1_p_max"] / g["p_in"][2], rtol=1e-8)


def test_sharded_sweep_through_rccl_with_one_rank(oracle):
    """The nccl (= RCCL) leg of distributed.sweep_sharded / DeviceSweep.gather on the one GPU this box has:
    world_size 1, rendezvous on 127.0.0.1.  (world_size 2 is covered on CPU with gloo.)"""
    import socket
    import torch
    import torch.distributed as dist
    from psa_amd.distributed import DeviceSweep, sweep_sharded, unpack_gathered
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        db = np.linspace(-0.05, 0.05, 333)
        a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
        ref = oracle.sweep(db, z_max=50.0, n=500, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        res = sweep_sharded(db, n_steps=500, z_max=50.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        assert rel_err(res.a_end, ref["a_end"]) < RTOL_F64 and np.array_equal(res.first_bad_step, ref["first_bad_step"])
        ds = DeviceSweep(db, n_steps=500, z_max=50.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        ds.launch()
        g = ds.gather()
        torch.cuda.synchronize()
        a, pe, pm, fb = unpack_gathered(ds.layout, g.cpu().numpy(), db.size, 1)
        assert np.array_equal(a, res.a_end) and np.array_equal(pm, res.p_max) and (fb == -1).all()
    finally:
        dist.destroy_process_group()


def test_grid_scan_rows_equal_the_1d_driver(golden):
    """scan_gain_grid (config 3's 2-D shape, here 5 x 16): every row must equal the reference-shaped 1-D driver."""
    g = golden("G11")
    dv = g["disp_m"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    cfg = config.custom_simulation_config(z_max=300.0, dz=0.25, save_every=7)
    lam2 = np.array([1556e-9, 1553e-9, 1558e-9, 0.5e-6, 1560e-9])     # incl. a far-detuned (but valid) pump 2
    kw = dict(cfg=cfg, lambda_p1_m=1550e-9, gamma=0.0115, alpha=1.0e-4, p_in=g["p_in"], phase_in=g["phase_in"],
              dispersion=d, length_unit="m", gain_unit="linear")
    grid = scan_mismtach.scan_gain_grid(lambda_p2_m=lam2, lambda_signal_m=g["lambda3"], **kw)
    assert grid["gain"].shape == (5, 16) and grid["dbeta"].shape == (5, 16)
    np.testing.assert_allclose(grid["gain"][0], g["m_gain"], rtol=RTOL_F64)       # row 0 is golden G11 itself
    assert np.array_equal(grid["dbeta"][0], g["m_dbeta"])
    for iy, l2 in enumerate(lam2):
        x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
            lambda_p2_m=float(l2), lambda_signal_m=g["lambda3"], return_wavelength_unit="m", show=False, **kw)
        assert np.array_equal(grid["gain"][iy], gain, equal_nan=True) and np.array_equal(grid["dbeta"][iy], db, equal_nan=True)
    iy, ix = grid["best_index"]
    assert grid["best_gain"] == np.nanmax(grid["gain"]) == grid["gain"][iy, ix]
    assert grid["n_finite"] == int(np.isfinite(grid["gain"]).sum())


def test_config3_full_grid_through_the_grid_driver(oracle):
    """BASELINE config 3 as SURVEY 8(d) defines it: 1024 x 1024 grid (pump-2 wavelength x signal wavelength),
    lambda_p1 = 1550 nm, dispersion D = 0.1 / S = 0.02, SYMMETRIC_EVEN (2, 4), P = (0.1, 0.1, 1e-7, 1e-7) W,
    L = 1000 m, 100 000 z-steps, float64 -- one launch of 1 048 576 points (~1 s).  Sampled points against the oracle."""
    lam2 = np.linspace(1552e-9, 1562e-9, 1024)
    lam3 = np.linspace(1540e-9, 1565e-9, 1024)
    om = frequency_plan.plan_from_wavelengths(1550e-9, 1558e-9, 1540e-9)
    sp = frequency_plan.infer_symmetry_from_omegas(*om)
    d = dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.1, 0.02, 0,
                                              D_units="ps/nm/km", S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                              omega_ref=sp.omega_c)
    p_in = np.array([0.1, 0.1, 1e-7, 1e-7])
    cfg = config.custom_simulation_config(z_max=1000.0, dz=0.01)
    grid = scan_mismtach.scan_gain_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=lam2, lambda_signal_m=lam3, gamma=0.0115,
                                        alpha=1.15e-4, p_in=p_in, dispersion=d, gain_unit="linear")
    assert grid["gain"].shape == (1024, 1024) and grid["n_finite"] == 1024 * 1024
    res = grid["result"]
    assert res.n_steps == 100_000 and (res.first_bad_step == -1).all()
    rng = np.random.default_rng(33)
    pick = rng.choice(1024 * 1024, 24, replace=False)
    db = grid["dbeta"].reshape(-1)[pick]
    ref = oracle.sweep(db, z_max=1000.0, n=100_000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=np.sqrt(p_in).astype(complex))
    assert rel_err(res.a_end[pick], ref["a_end"]) < RTOL_F64
    np.testing.assert_allclose(grid["gain"].reshape(-1)[pick], ref["p_max"] / p_in[2], rtol=RTOL_F64)
    iy, ix = grid["best_index"]
    assert grid["gain"][iy, ix] == grid["gain"].max() > 1.0


def test_the_ctypes_stub_printed_in_integration_md_runs(golden):
    """INTEGRATION.md shows the binding a maintainer of the reference would add (psa_hip.py).  Execute exactly that
    text (library path substituted) and check it against golden G8, so the document cannot rot."""
    import os
    import re
    import types
    import psa_amd._native as nat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    block = re.search(r"```python\n(# psa_hip.py.*?)```", doc, re.S).group(1)
    block = block.replace("/path/to/psa-simulation-ode-rk-mvp-dispersion_amd/libpsa_hip.so", nat.LIB_PATH)
    stub = types.ModuleType("psa_hip_stub")
    exec(compile(block, "INTEGRATION.md:psa_hip.py", "exec"), stub.__dict__)
    g = golden("G8")
    a_end, p_end, p_max, bad, traj = stub.rk4_sweep(g["dbeta257"], float(g["gamma"]), float(g["alphas"][1]),
                                                    np.sqrt(g["p_in"]).astype(complex), 1000.0, 10_000, 10)
    assert rel_err(a_end, g["n1e4_a1_A_end"]) < RTOL_F64 and rel_err(p_max, g["n1e4_a1_p_max"]) < RTOL_F64
    assert (bad == -1).all() and traj is None
    a1, _, _, b1, t1 = stub.rk4_sweep([0.013], 0.0115, 1.15e-4, np.sqrt(g["p_in"]).astype(complex), 100.5, 1005, 10, want_traj=True)
    assert t1.shape == (1, 101, 4) and np.array_equal(t1[0, -1], a1[0])
    # section B4 of the same document: the dbeta producer bound the same way, against the reference's own dbeta (G2)
    b4 = re.search(r"```python\n(_L\.psa_dbeta_grid_f64\.restype.*?)```", doc, re.S).group(1)
    from psa_amd import constants
    stub.constants = constants
    exec(compile(b4, "INTEGRATION.md:B4", "exec"), stub.__dict__)
    g2 = golden("G2")
    d = dispersion.DispersionParams(omega_ref=float(g2["omega_ref"]), beta2=float(g2["beta2"]), beta3=float(g2["beta3"]),
                                    beta4=float(g2["beta4"]))
    db, ok = stub.dbeta_grid(d, float(g2["lambda_p1"]), [float(g2["lambda_p2"])], g2["lambda3"])
    assert ok.all() and np.all(np.abs(db - g2["dbeta"]) <= np.spacing(np.abs(g2["dbeta"])))


def test_device_entry_point_can_be_captured_into_a_graph_and_replayed(oracle):
    """include/psa_rk4.h: the `_dev` entry points neither allocate nor synchronise.  Capture one sweep launch into a
    HIP graph (through torch.cuda.CUDAGraph), change the inputs in place, replay, and compare with the oracle."""
    import torch
    from psa_amd.distributed import DeviceSweep
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    db1 = np.linspace(-0.05, 0.05, 500)
    db2 = np.linspace(0.02, -0.03, 500)
    ds = DeviceSweep(db1, n_steps=800, z_max=80.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0,
                     device=torch.device("cuda", 0))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ds.launch()                                   # warm-up outside capture (module load)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ds.launch()
    for db in (db1, db2):
        ds.dbeta.copy_(torch.as_tensor(db, device=ds.device))
        ds.record.zero_()
        graph.replay()
        torch.cuda.synchronize()
        a, pe, pm, fb = ds.layout.unpack(ds.record.cpu().numpy(), db.size)
        ref = oracle.sweep(db, z_max=80.0, n=800, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        assert rel_err(a, ref["a_end"]) < RTOL_F64 and rel_err(pm, ref["p_max"]) < RTOL_F64 and (fb == -1).all()


def test_plain_c_program_drives_a_sweep_through_the_abi(tmp_path):
    """tests/c/abi_gpu_client.c: a C99 program (no Python, no torch) runs a 257-point sweep with trajectories, the gain
    summary and one RHS evaluation through libpsa_hip.so and checks A[-1] == last saved row."""
    import os
    import shutil
    import subprocess
    import psa_amd._native as nat
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(nat.LIB_PATH)
    exe = str(tmp_path / "abi_gpu_client")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "abi_gpu_client.c"), "-o", exe, "-L", libdir, "-lpsa_hip", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "abi_gpu_client ok" in out.stdout


def _two_rank_native_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psa_amd.distributed import sweep_sharded
        rng = np.random.default_rng(3)
        db = rng.uniform(-0.05, 0.05, 1001)                     # ragged split: 501 + 500
        gam = rng.uniform(5e-3, 2e-2, 1001)
        a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
        res = sweep_sharded(db, n_steps=2000, z_max=200.0, save_every=10, gamma=gam, alpha=1.15e-4, a0=a0, device=0)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), a_end=res.a_end, p_max=res.p_max, bad=res.first_bad_step, db=db, gam=gam)
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_native_executor_gloo_gather(tmp_path, oracle):
    """N > 1 end to end with the REAL kernel on the one GPU this box has: two processes (well inside the 6-process
    limit), each integrates its shard on device 0 through libpsa_hip.so, `gloo` carries the single gather; both ranks
    must hold the full result and it must match the oracle.  (RCCL needs one GPU per rank: that leg runs with one rank,
    see test_sharded_sweep_through_rccl_with_one_rank; the 8-GPU run is the driver's.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_native_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in ("a_end", "p_max", "bad"):
        assert np.array_equal(r0[k], r1[k])
    ref = oracle.sweep(r0["db"], z_max=200.0, n=2000, save_every=10, gamma=r0["gam"], alpha=1.15e-4,
                       a0=np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex))
    assert r0["a_end"].shape == (1001, 4) and rel_err(r0["a_end"], ref["a_end"]) < RTOL_F64
    assert rel_err(r0["p_max"], ref["p_max"]) < RTOL_F64 and (r0["bad"] == -1).all()


def test_six_wave_grid_driver(golden, oracle):
    """scan_six_wave_grid (config 5's shape, here 6 x 9): sampled points against the oracle's 6-wave statement, and the
    reduction property -- with pair 2 dark, every column equals the 4-wave sweep at dbeta1."""
    g = golden("G11")
    dv = g["disp_m"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    cfg = config.custom_simulation_config(z_max=300.0, dz=0.1)
    O1 = np.linspace(2e12, 9e12, 6)
    O2 = np.linspace(-8e12, -1e12, 9)
    P6 = np.array([0.3, 0.25, 2e-6, 5e-7, 1e-6, 1e-6])
    out = scan_mismtach.scan_six_wave_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1556e-9, Omega1=O1, Omega2=O2,
                                           gamma=0.0115, alpha=1.0e-4, p_in=P6, dispersion=d, gain_unit="linear")
    assert out["gain"].shape == (6, 9) and out["a_end"].shape == (6, 9, 6) and (out["first_bad_step"] == -1).all()
    a06 = np.sqrt(P6).astype(complex)
    for iy, ix in ((0, 0), (2, 5), (5, 8)):
        ref = oracle.sweep(np.array([out["dbeta1"][iy]]), z_max=300.0, n=3000, save_every=10, gamma=0.0115, alpha=1.0e-4,
                           a0=a06, dbeta2=np.array([out["dbeta2"][ix]]))
        assert rel_err(out["a_end"][iy, ix], ref["a_end"][0]) < RTOL_F64
        assert out["gain"][iy, ix] == pytest.approx(ref["p_max"][0] / P6[2], rel=RTOL_F64)
    dark = P6.copy()
    dark[4:] = 0.0
    o2 = scan_mismtach.scan_six_wave_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1556e-9, Omega1=O1, Omega2=O2,
                                          gamma=0.0115, alpha=1.0e-4, p_in=dark, dispersion=d, gain_unit="linear")
    r4 = scan_mismtach.scan_dbeta_seeded_signal(cfg=cfg, delta_beta=o2["dbeta1"], gamma=0.0115, alpha=1.0e-4, p_in=dark[:4],
                                                gain_mode="max", gain_unit="linear")
    for ix in range(9):
        np.testing.assert_allclose(o2["gain"][:, ix], r4["gain"], rtol=1e-11)
    assert np.all(o2["a_end"][..., 4:] == 0)
    with pytest.raises(ValueError):
        scan_mismtach.scan_six_wave_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1556e-9, Omega1=O1, Omega2=O2,
                                         gamma=0.0115, alpha=0.0, p_in=P6[:4], dispersion=d)


def test_concurrent_host_api_calls_from_four_threads():
    """include/psa_rk4.h: the host-buffer entry points keep no global state (own stream + own device buffers per call).
    Four threads sweep different inputs at the same time (ctypes drops the GIL); every result must equal the
    sequential one bit for bit, and per-thread error strings must not leak between threads."""
    import threading
    import psa_amd._native as nat
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    jobs = [dict(dbeta=np.linspace(-0.05, 0.05, 3000 + 17 * k), n_steps=4000 + 100 * k, z_max=400.0 + k, save_every=10,
                 gamma=0.0115 + 1e-4 * k, alpha=1.15e-4) for k in range(4)]
    run = lambda j: nat.sweep_host(j["dbeta"], n_steps=j["n_steps"], z_max=j["z_max"], save_every=j["save_every"],  # noqa: E731
                                   gamma=j["gamma"], alpha=j["alpha"], a0=a0)
    sequential = [run(j) for j in jobs]
    results, errors = [None] * 4, []

    def worker(k):
        try:
            for _ in range(3):
                results[k] = run(jobs[k])
            with pytest.raises(nat.PsaNativeError) as e:       # an argument error on this thread only
                nat.sweep_host(np.zeros(3), n_steps=0, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=a0)
            assert e.value.code == -3
        except BaseException as exc:  # noqa: BLE001
            errors.append((k, exc))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for seq, par in zip(sequential, results):
        assert np.array_equal(seq["a_end"], par["a_end"]) and np.array_equal(seq["p_max"], par["p_max"])
        assert np.array_equal(seq["first_bad_step"], par["first_bad_step"])


def test_g13_reference_runs_through_the_legacy_and_taylor_paths(golden):
    """G13 (generated from the reference): run_single_simulation with only legacy betas (m and km: the reference scales the
    legacy dbeta twice in km), with GENERAL_TAYLOR, and the 4 x 9 grid through scan_gain_grid with the host AND the device
    dbeta producer, default and GENERAL_TAYLOR phase matching -- all against the reference's own numbers."""
    g = golden("G13")
    kw = dict(omega=g["omega"], p_in=g["p_in"], phase_in=g["phase_in"])
    cfg_m = config.custom_simulation_config(z_max=200.0, dz=0.2, save_every=8)
    z, A = simulation.run_single_simulation(cfg_m, gamma=0.0115, alpha=1.0e-4, beta_legacy=g["beta_legacy_m"], **kw)
    assert rel_err(A, g["legacy_m_A"]) < RTOL_F64 and np.allclose(z, g["legacy_m_z"], rtol=1e-15)
    cfg_km = config.custom_simulation_config(z_max=0.2, dz=0.2e-3, save_every=8)
    z, A = simulation.run_single_simulation(cfg_km, gamma=11.5, alpha=0.1, beta_legacy=g["beta_legacy_m"] * 1e3,
                                            length_unit="km", return_length_unit="m", **kw)
    assert rel_err(A, g["legacy_km_A"]) < RTOL_F64 and np.allclose(z, g["legacy_km_z"], rtol=1e-13)
    dv = g["disp"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    gen = PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4)
    z, A = simulation.run_single_simulation(cfg_m, gamma=0.0115, alpha=1.0e-4, dispersion=d, phase_matching_cfg=gen, **kw)
    assert rel_err(A, g["gen_A"]) < RTOL_F64
    cfg_g = config.custom_simulation_config(z_max=250.0, dz=0.25, save_every=5)
    for tag, pm in (("sym", None), ("gen", gen)):
        for producer in ("host", "device"):
            out = scan_mismtach.scan_gain_grid(cfg=cfg_g, lambda_p1_m=1550e-9, lambda_p2_m=g["lambda2"],
                                               lambda_signal_m=g["lambda3"], gamma=0.0115, alpha=1.0e-4, p_in=g["grid_p_in"],
                                               dispersion=d, phase_matching_cfg=pm, dbeta_producer=producer)
            assert np.max(np.abs(out["gain"] - g["grid_gain_" + tag])) < ATOL_DB, (tag, producer)
            if tag == "sym":
                assert np.all(np.abs(out["dbeta"] - g["grid_dbeta_sym"]) <= np.spacing(np.abs(g["grid_dbeta_sym"])))
            else:
                np.testing.assert_allclose(out["dbeta"], g["grid_dbeta_gen"], rtol=1e-14, atol=0)
