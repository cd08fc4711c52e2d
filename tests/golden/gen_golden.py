#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the reference.

Runs ONLY in the build container, where the upstream reference is mounted
read-only at /root/reference.  Nothing from the reference is copied: the
outputs are data (inputs -> expected outputs) stored as .npz without pickles.
The GPU box never sees the reference; tests there read these fixtures.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [--only G1,G8]

Fixture ids follow SURVEY.md section 8(c):
  G1  main.py single-point scenario (10 000 steps): omega, beta_n, dbeta, full A
  G2  30-point lambda3 sweep (gain + dbeta driver) + per-point A_end
  G3  100-point lambda3 sweep (gain driver), 45 dB peak
  G4  simulation.example_zero_signal / custom_seeded_signal (km unit path)
  G5  64 random direct RHS evaluations + the three RHS terms
  G6  generic stepper known-answer cases of the reference's passing tests
  G7  save-stride / non-integer z_max/dz edge cases
  G8  direct PROVIDED-dbeta sweeps (main kernel-parity fixture)
  G9  failure path: FloatingPointError step index, NaN mask through the driver
  G10 dbeta producer cases (unit variants, GENERAL_TAYLOR, dS/dlambda != 0 quirk)
  G11 driver variants: km units, linear gain, non-zero input phases
  G12 a run bundle written by the reference's io_fwm (npz + csv + json): pins the file format
  API api_signatures.json: parameter names / kinds / defaults of every public function and the field names of every public
      dataclass of the hot-path modules (the drop-in boundary as data)
  G15 SURVEY 8(d)'s robustness draw through the reference: 32 single runs with every physical and numerical parameter random
  ERR error_contract.json: exception type + message the reference raises for the invalid calls listed in error_cases.py
  HOST host_api_records.json.gz: every return value (or exception type) of the scalar host API along the seeded random walk
      of host_api_cases.py (300 plans x dispersion builders x phase-matching methods, 40 configurations)
  G13 paths the other fixtures do not walk: legacy beta(w_j) fallback (m and km), GENERAL_TAYLOR through the single run and
      through the gain+dbeta driver, and a 4 x 9 (lambda_p2 x lambda_signal) grid run row by row through the reference's
      driver (pins the build's 2-D grid scan and its device-side dbeta producer against the reference itself)
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from multiprocessing import Pool

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = os.environ.get("PSA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _save(name: str, **arrays) -> None:
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path)} B)", flush=True)


# --------------------------------------------------------------------------
def _g1_inputs():
    import frequency_plan as fp
    import dispersion as dp
    lam = (1550e-9, 1560e-9, 1555e-9)
    omega = fp.plan_from_wavelengths(*lam, lambda4_m=None)
    sp = fp.infer_symmetry_from_omegas(omega[0], omega[1], omega[2], omega[3])
    lam_c = fp.lambda_from_omega(sp.omega_c)
    disp = dp.dispersion_params_from_D_S(
        lambda_ref_m=lam_c, D=0.02, S=0.02, dSdlmbd=0, D_units="ps/nm/km",
        S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)
    return lam, omega, sp, lam_c, disp


def gen_g1():
    import config, dispersion as dp, simulation
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    lam, omega, sp, lam_c, disp = _g1_inputs()
    cfg = config.custom_simulation_config(z_max=1000.0, dz=0.1)
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, even_orders=(2, 4), max_order=4)
    gamma = 11.5 / 1000.0
    alpha = (np.log(10.0) / 10.0) * 0.9 / 1000.0
    p_in = np.array([0.5, 0.5, 1e-5, 1e-5])
    z, A = simulation.run_single_simulation(
        cfg, gamma=gamma, alpha=alpha, omega=omega, p_in=p_in, phase_in=np.zeros(4),
        dispersion=disp, phase_matching_cfg=pm, length_unit="m", return_length_unit="m")
    gain_db = 10.0 * np.log10(np.abs(A[-1, 2]) ** 2 / p_in[2])
    _save("G1", lam=np.array(lam), omega=omega, omega_c=sp.omega_c, omega_d=sp.omega_d, Omega=sp.Omega,
          lambda_c=lam_c, beta2=disp.beta2, beta3=disp.beta3, beta4=disp.beta4,
          dbeta_sym=dp.delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, disp),
          dbeta_gen=dp.delta_beta_from_omegas(omega, disp),
          gamma=gamma, alpha=alpha, p_in=p_in, z_max=1000.0, dz=0.1, save_every=10,
          z=z, A=A, gain_db=gain_db)


def _sweep_setup(lp1, lp2, lam3, D):
    import frequency_plan as fp
    import dispersion as dp
    om = fp.plan_from_wavelengths(lp1, lp2, float(lam3[0]), lambda4_m=None)
    sp = fp.infer_symmetry_from_omegas(om[0], om[1], om[2], om[3])
    lam_c = fp.lambda_from_omega(sp.omega_c)
    disp = dp.dispersion_params_from_D_S(
        lambda_ref_m=lam_c, D=D, S=0.02, dSdlmbd=0, D_units="ps/nm/km",
        S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)
    return sp, lam_c, disp


def _a_end_point(args):
    (lp1, lp2, l3, gamma, alpha, p_in, z_max, dz, disp_tuple) = args
    import config, simulation, frequency_plan as fp
    from dispersion import DispersionParams
    disp = DispersionParams(omega_ref=disp_tuple[0], beta2=disp_tuple[1], beta3=disp_tuple[2], beta4=disp_tuple[3])
    cfg = config.custom_simulation_config(z_max=z_max, dz=dz)
    om = fp.plan_from_wavelengths(lp1, lp2, l3, lambda4_m=None)
    z, A = simulation.run_single_simulation(cfg, gamma=gamma, alpha=alpha, omega=om, p_in=p_in,
                                            phase_in=np.zeros(4), dispersion=disp)
    return A[-1], float(np.max(np.abs(A[:, 2]) ** 2))


def gen_g2(pool):
    import config, scan_mismtach as sm
    lp1, lp2 = 1550e-9, 1558e-9
    lam3 = np.linspace(1540e-9, 1565e-9, 30)
    sp, lam_c, disp = _sweep_setup(lp1, lp2, lam3, 0.1)
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    gamma = 11.5 / 1000.0
    alpha = (np.log(10.0) / 10.0) * 0.5 / 1000.0
    p_in = np.array([0.1, 0.1, 1e-7, 1e-7])
    x, g, db = sm.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=lp1, lambda_p2_m=lp2, lambda_signal_m=lam3, gamma=gamma, alpha=alpha,
        p_in=p_in, dispersion=disp, length_unit="m", gain_unit="dB", phase_in=np.zeros(4),
        show=False, show_progress=False)
    dt = (disp.omega_ref, disp.beta2, disp.beta3, disp.beta4)
    res = pool.map(_a_end_point, [(lp1, lp2, float(l), gamma, alpha, p_in, 500.0, 0.2, dt) for l in lam3])
    _save("G2", lambda_p1=lp1, lambda_p2=lp2, lambda3=lam3, D=0.1, S=0.02, lambda_c=lam_c,
          omega_ref=disp.omega_ref, beta2=disp.beta2, beta3=disp.beta3, beta4=disp.beta4,
          gamma=gamma, alpha=alpha, p_in=p_in, z_max=500.0, dz=0.2, save_every=10,
          x=x, gain_db=g, dbeta=db, A_end=np.array([r[0] for r in res]), p3_max=np.array([r[1] for r in res]))


def _g3_chunk(args):
    import config, scan_mismtach as sm
    from dispersion import DispersionParams
    (lp1, lp2, lam3, gamma, alpha, p_in, dt, unit) = args
    disp = DispersionParams(omega_ref=dt[0], beta2=dt[1], beta3=dt[2], beta4=dt[3])
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    x, g = sm.plot_max_signal_gain_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=lp1, lambda_p2_m=lp2, lambda_signal_m=lam3, gamma=gamma, alpha=alpha,
        p_in=p_in, phase_in=np.zeros(4), dispersion=disp, length_unit="m", return_wavelength_unit="nm",
        gain_unit=unit, show=False, show_progress=False)
    return x, g


def gen_g3(pool):
    from phase_matching import PhaseMatchingConfig, compute_phase_mismatch
    import frequency_plan as fp
    lp1, lp2 = 1550e-9, 1555e-9
    lam3 = np.linspace(1540e-9, 1650e-9, 100)
    sp, lam_c, disp = _sweep_setup(lp1, lp2, lam3, 0.2)
    gamma = 11.5 / 1000.0
    alpha = (np.log(10.0) / 10.0) * 0.5 / 1000.0
    p_in = np.array([0.5, 0.5, 1e-7, 1e-7])
    dt = (disp.omega_ref, disp.beta2, disp.beta3, disp.beta4)
    chunks = np.array_split(lam3, 10)
    out = pool.map(_g3_chunk, [(lp1, lp2, c, gamma, alpha, p_in, dt, "db") for c in chunks])
    x = np.concatenate([o[0] for o in out])
    g = np.concatenate([o[1] for o in out])
    pm = PhaseMatchingConfig()
    db = np.array([compute_phase_mismatch(fp.plan_from_wavelengths(lp1, lp2, float(l)), disp, pm).delta_beta
                   for l in lam3])
    _save("G3", lambda_p1=lp1, lambda_p2=lp2, lambda3=lam3, D=0.2, S=0.02, lambda_c=lam_c,
          omega_ref=disp.omega_ref, beta2=disp.beta2, beta3=disp.beta3, beta4=disp.beta4,
          gamma=gamma, alpha=alpha, p_in=p_in, z_max=500.0, dz=0.2, save_every=10,
          x=x, gain_db=g, dbeta=db)


def gen_g4():
    import simulation
    z1, A1 = simulation.example_zero_signal()
    z2, A2 = simulation.custom_seeded_signal()
    _save("G4", zero_z=z1, zero_A=A1, seeded_z=z2, seeded_A=A2)


def gen_g5():
    import yaman_model as ym
    from parameters import FiberParams, SimulationGrid, WavesParams, PhaseMatchingParams, make_model_params
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    rng = np.random.default_rng(12345)
    n = 64
    z = rng.uniform(0.0, 1000.0, n)
    a = (rng.normal(size=(n, 4)) + 1j * rng.normal(size=(n, 4))) * rng.uniform(1e-3, 1.0, (n, 1))
    gamma = rng.uniform(5e-3, 2e-2, n)
    alpha = rng.uniform(0.0, 3e-4, n)
    alpha[::8] = 0.0
    dbeta = rng.uniform(-0.1, 0.1, n)
    rhs = np.empty((n, 4), complex); lin = np.empty_like(rhs); kerr = np.empty_like(rhs); fwm = np.empty_like(rhs)
    w0 = 2 * np.pi * 299792458.0 / 1.55e-6
    for i in range(n):
        pm = PhaseMatchingParams(config=PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED,
                                                             provided_delta_beta=float(dbeta[i])))
        params = make_model_params(
            waves=WavesParams(omega=np.full(4, w0)),
            fiber=FiberParams(length_m=1000.0, gamma_W_m=float(gamma[i]), alpha_1_m=float(alpha[i])),
            grid=SimulationGrid(dz_m=0.1), phase_matching=pm)
        params.cache.set_phase_mismatch(float(dbeta[i]))
        rhs[i] = ym.rhs_yaman_simplified(float(z[i]), a[i], params)
        lin[i] = ym._linear_loss_terms(a[i], float(alpha[i]))
        kerr[i] = ym._kerr_terms(a[i], float(gamma[i]))
        fwm[i] = ym._fwm_terms(float(z[i]), a[i], float(gamma[i]), float(dbeta[i]))
    _save("G5", z=z, a=a, gamma=gamma, alpha=alpha, dbeta=dbeta, rhs=rhs, linear=lin, kerr=kerr, fwm=fwm)


def gen_g6():
    import integrators
    f = lambda z, y, p: y  # noqa: E731
    y1 = integrators.rk4_step(f, 0.0, np.array([1.0]), 0.1, None)
    z_out, y_out = integrators.integrate_interval(f, 1.0, 0.1, np.array([1.0]), None, save_every=2, check_nan=True)
    # a 2-state complex linear system too (rotation + decay), exercises dtype passthrough
    M = np.array([[-0.1 + 2.0j, 0.3], [-0.3, -0.2 - 1.0j]])
    g = lambda z, y, p: M @ y * (1.0 + 0.1 * z)  # noqa: E731
    z2, y2 = integrators.integrate_interval(g, 2.0, 0.01, np.array([1.0 + 0j, 0.5j]), None, save_every=7)
    _save("G6", rk4_step_exp=y1, interval_z=z_out, interval_y=y_out, lin_M=M, lin_z=z2, lin_y=y2)


def _provided_run(z_max, dz, save_every, dbeta, gamma, alpha, p_in, phase=None, check_nan=True):
    import config, simulation
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    cfg = config.custom_simulation_config(z_max=z_max, dz=dz, save_every=save_every, check_nan=check_nan)
    w0 = 2 * np.pi * 299792458.0 / 1.55e-6
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=float(dbeta))
    return simulation.run_single_simulation(cfg, gamma=gamma, alpha=alpha, omega=np.full(4, w0), p_in=p_in,
                                            phase_in=phase, phase_matching_cfg=pm)


def gen_g7():
    p_in = np.array([0.5, 0.5, 1e-5, 1e-5])
    out = {}
    for tag, (z_max, dz, se) in {"n1005_se10": (100.5, 0.1, 10), "n1005_se1": (100.5, 0.1, 1),
                                 "n3_se1": (1.0, 0.3, 1), "n3_se2": (1.0, 0.3, 2),
                                 "n7_se10": (0.7, 0.1, 10)}.items():
        z, A = _provided_run(z_max, dz, se, 0.013, 0.0115, 1.15e-4, p_in, phase=np.array([0.1, -0.2, 0.3, 0.7]))
        out[tag + "_z"] = z
        out[tag + "_A"] = A
        out[tag + "_cfg"] = np.array([z_max, dz, se])
    _save("G7", dbeta=0.013, gamma=0.0115, alpha=1.15e-4, p_in=p_in, phase_in=np.array([0.1, -0.2, 0.3, 0.7]), **out)


def _g8_point(args):
    (dbeta, alpha, n) = args
    p_in = np.array([0.5, 0.5, 1e-5, 1e-5])
    z, A = _provided_run(1000.0, 1000.0 / n, 10, dbeta, 0.0115, alpha, p_in)
    P3 = np.abs(A[:, 2]) ** 2
    return A[-1], float(P3[-1]), float(P3.max())


def gen_g8(pool):
    out = {}
    db257 = np.linspace(-0.05, 0.05, 257)
    for tag, alpha in (("a0", 0.0), ("a1", 1.15e-4)):
        res = pool.map(_g8_point, [(float(d), alpha, 10_000) for d in db257], chunksize=4)
        out[f"n1e4_{tag}_A_end"] = np.array([r[0] for r in res])
        out[f"n1e4_{tag}_p_end"] = np.array([r[1] for r in res])
        out[f"n1e4_{tag}_p_max"] = np.array([r[2] for r in res])
        print(f"  G8 n=1e4 {tag} done", flush=True)
    db33 = np.linspace(-0.05, 0.05, 33)
    res = pool.map(_g8_point, [(float(d), 1.15e-4, 100_000) for d in db33], chunksize=1)
    out["n1e5_a1_A_end"] = np.array([r[0] for r in res])
    out["n1e5_a1_p_end"] = np.array([r[1] for r in res])
    out["n1e5_a1_p_max"] = np.array([r[2] for r in res])
    _save("G8", dbeta257=db257, dbeta33=db33, gamma=0.0115, alphas=np.array([0.0, 1.15e-4]),
          p_in=np.array([0.5, 0.5, 1e-5, 1e-5]), z_max=1000.0, save_every=10, **out)


def gen_g9():
    import config, scan_mismtach as sm
    p_in = np.array([0.5, 0.5, 1e-5, 1e-5])
    steps = []
    # 12.0 .. 10.29612 sit just past the RK4 stability edge (gamma*P*h ~ 1): the first bad step moves 2..5;
    # 10.0 is stable (-1)
    gammas = np.array([50.0, 200.0, 1e3, 12.0, 10.29664, 10.29613, 10.29612, 10.0])
    for g in gammas:
        try:
            _provided_run(100.0, 0.1, 10, 0.01, float(g), 0.0, p_in)
            steps.append(-1)
        except FloatingPointError as e:
            msg = str(e)  # "NaN or Inf detected at step {i}, z = {z}"
            steps.append(int(msg.split("step")[1].split(",")[0]))
    # same inputs with check_nan=False: NaNs stored silently
    z, A = _provided_run(100.0, 0.1, 10, 0.01, 200.0, 0.0, p_in, check_nan=False)
    first_bad_row = int(np.argmax(~np.isfinite(A).all(axis=1)))
    # NaN mask through the sweep driver (overflowing gamma at every point -> all NaN; sane gamma -> finite)
    lam3 = np.linspace(1545e-9, 1560e-9, 6)
    sp, lam_c, disp = _sweep_setup(1550e-9, 1558e-9, lam3, 0.1)
    cfg = config.custom_simulation_config(z_max=100.0, dz=0.1)
    x, g_bad = sm.plot_max_signal_gain_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3, gamma=200.0, alpha=0.0,
        p_in=p_in, dispersion=disp, show=False, show_progress=False)
    # one impossible wavelength (omega4 <= 0) inside an otherwise fine sweep -> NaN only there
    lam3b = np.array([1550e-9, 0.7e-6, 1556e-9])
    xb, g_mixed, db_mixed = sm.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3b, gamma=0.0115, alpha=0.0,
        p_in=p_in, dispersion=disp, show=False, show_progress=False)
    _save("G9", gammas=gammas, first_bad_step=np.array(steps), p_in=p_in, dbeta=0.01, z_max=100.0, dz=0.1,
          nocheck_first_bad_row=first_bad_row, nocheck_n_rows=A.shape[0],
          drv_lambda3=lam3, drv_gain_bad=g_bad, drv_omega_ref=disp.omega_ref, drv_beta2=disp.beta2,
          drv_beta3=disp.beta3, drv_beta4=disp.beta4,
          mixed_lambda3=lam3b, mixed_gain=g_mixed, mixed_dbeta=db_mixed)


def gen_g10():
    import dispersion as dp, frequency_plan as fp
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod, compute_phase_mismatch
    rows = []
    # dispersion_params_from_D_S variants incl. dS/dlambda != 0 (pins the beta4 argument-slot quirk)
    for (lam, D, S, dS, du, su, dsu) in [
        (1554.9e-9, 0.02, 0.02, 0.0, "ps/nm/km", "ps/nm^2/km", "ps/nm^3/km"),
        (1550e-9, 0.5, 0.07, 1e-4, "ps/nm/km", "ps/nm^2/km", "ps/nm^3/km"),
        (1550e-9, 1e-6, 50.0, 2e8, "SI", "SI", "SI"),
        (1310e-9, -1.5, None, None, "ps/nm/km", "ps/nm^2/km", "ps/nm^3/km"),
        (1565e-9, 3.0, 0.08, None, "ps/nm/km", "ps/nm^2/km", "ps/nm^3/km"),
    ]:
        d = dp.dispersion_params_from_D_S(lam, D, S, dS, D_units=du, S_units=su, dSdlmbd_units=dsu)
        rows.append([lam, D, np.nan if S is None else S, np.nan if dS is None else dS,
                     {"SI": 0, "ps/nm/km": 1}[du], d.omega_ref, d.beta2, d.beta3, d.beta4])
    rng = np.random.default_rng(7)
    lp1 = rng.uniform(1540e-9, 1560e-9, 40)
    lp2 = rng.uniform(1545e-9, 1570e-9, 40)
    l3 = rng.uniform(1500e-9, 1620e-9, 40)
    disp = dp.DispersionParams(omega_ref=fp.omega_from_lambda(1552e-9), beta2=-2.3e-28, beta3=4.1e-41, beta4=-3.0e-55,
                               extra={6: 1.0e-84})
    om = np.array([fp.plan_from_wavelengths(a, b, c) for a, b, c in zip(lp1, lp2, l3)])
    sym = np.array([[s.omega_c, s.omega_d, s.Omega] for s in
                    (fp.infer_symmetry_from_omegas(*o) for o in om)])
    cfgs = {
        "sym24": PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, even_orders=(2, 4)),
        "sym2": PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, even_orders=(2,)),
        "sym246": PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, even_orders=(2, 4, 6)),
        "gen4": PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4),
        "gen2": PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=2),
        "gen6": PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=6),
    }
    out = {k: np.array([compute_phase_mismatch(o, disp, c).delta_beta for o in om]) for k, c in cfgs.items()}
    _save("G10", ds_rows=np.array(rows, dtype=float), lp1=lp1, lp2=lp2, l3=l3, omega=om, sym=sym,
          disp=np.array([disp.omega_ref, disp.beta2, disp.beta3, disp.beta4, 1.0e-84]), **out)


def _g11_chunk(args):
    import config, scan_mismtach as sm
    from dispersion import DispersionParams
    (lam3, dt, kw) = args
    disp = DispersionParams(omega_ref=dt[0], beta2=dt[1], beta3=dt[2], beta4=dt[3])
    cfg = config.custom_simulation_config(z_max=kw["z_max"], dz=kw["dz"], save_every=kw["se"])
    x, g, db = sm.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1556e-9, lambda_signal_m=lam3, gamma=kw["gamma"],
        alpha=kw["alpha"], p_in=kw["p_in"], phase_in=kw["phase"], dispersion=disp, length_unit=kw["unit"],
        return_wavelength_unit="m", gain_unit="linear", show=False, show_progress=False)
    return x, g, db


def gen_g11(pool):
    lam3 = np.linspace(1546e-9, 1560e-9, 16)
    sp, lam_c, disp_m = _sweep_setup(1550e-9, 1556e-9, lam3, 0.15)
    p_in = np.array([0.3, 0.25, 2e-6, 5e-7])
    phase = np.array([0.3, -1.1, 2.0, 0.4])
    # km-unit run: all per-length quantities x1000, lengths /1000 -> must equal the m-unit run
    dt_m = (disp_m.omega_ref, disp_m.beta2, disp_m.beta3, disp_m.beta4)
    dt_km = (disp_m.omega_ref, disp_m.beta2 * 1e3, disp_m.beta3 * 1e3, disp_m.beta4 * 1e3)
    kw_m = dict(z_max=300.0, dz=0.25, se=7, gamma=0.0115, alpha=1.0e-4, p_in=p_in, phase=phase, unit="m")
    kw_km = dict(z_max=0.3, dz=0.25e-3, se=7, gamma=11.5, alpha=0.1, p_in=p_in, phase=phase, unit="km")
    ch = np.array_split(lam3, 4)
    rm = pool.map(_g11_chunk, [(c, dt_m, kw_m) for c in ch])
    rk = pool.map(_g11_chunk, [(c, dt_km, kw_km) for c in ch])
    _save("G11", lambda3=lam3, disp_m=np.array(dt_m), disp_km=np.array(dt_km), p_in=p_in, phase_in=phase,
          m_cfg=np.array([300.0, 0.25, 7, 0.0115, 1.0e-4]), km_cfg=np.array([0.3, 0.25e-3, 7, 11.5, 0.1]),
          m_x=np.concatenate([r[0] for r in rm]), m_gain=np.concatenate([r[1] for r in rm]),
          m_dbeta=np.concatenate([r[2] for r in rm]),
          km_x=np.concatenate([r[0] for r in rk]), km_gain=np.concatenate([r[1] for r in rk]),
          km_dbeta=np.concatenate([r[2] for r in rk]))


def _g13_row(args):
    import config, scan_mismtach as sm
    from dispersion import DispersionParams
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    (lp2, lam3, dt, method) = args
    disp = DispersionParams(omega_ref=dt[0], beta2=dt[1], beta3=dt[2], beta4=dt[3])
    cfg = config.custom_simulation_config(z_max=250.0, dz=0.25, save_every=5)
    pm = None if method == "sym" else PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4)
    x, g, db = sm.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=lp2, lambda_signal_m=lam3, gamma=0.0115, alpha=1.0e-4,
        p_in=np.array([0.4, 0.3, 1e-6, 1e-6]), phase_in=None, dispersion=disp, phase_matching_cfg=pm, length_unit="m",
        return_wavelength_unit="m", gain_unit="dB", show=False, show_progress=False)
    return g, db


def gen_g13(pool):
    import config, simulation, frequency_plan as fp
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    lam3 = np.linspace(1541e-9, 1566e-9, 9)
    lam2 = np.linspace(1553e-9, 1561e-9, 4)
    sp, lam_c, disp = _sweep_setup(1550e-9, 1557e-9, lam3, 0.12)
    dt = (disp.omega_ref, disp.beta2, disp.beta3, disp.beta4)
    # (a) legacy betas only: default method becomes PROVIDED with dbeta = b3 + b4 - b1 - b2 (simulation.py, yaman_model.py:112)
    om = fp.plan_from_wavelengths(1550e-9, 1557e-9, 1545e-9)
    p_in = np.array([0.4, 0.3, 1e-6, 2e-6])
    phase = np.array([0.0, 0.7, -0.4, 1.9])
    b_m = np.array([5.80e6 + 1.0e-3, 5.79e6 - 2.0e-3, 5.81e6 + 4.0e-3, 5.78e6 + 6.0e-3])         # 1/m -> dbeta = 0.011
    cfg_m = config.custom_simulation_config(z_max=200.0, dz=0.2, save_every=8)
    z_m, A_m = simulation.run_single_simulation(cfg_m, gamma=0.0115, alpha=1.0e-4, omega=om, p_in=p_in, phase_in=phase,
                                                beta_legacy=b_m)
    cfg_km = config.custom_simulation_config(z_max=0.2, dz=0.2e-3, save_every=8)
    z_km, A_km = simulation.run_single_simulation(cfg_km, gamma=11.5, alpha=0.1, omega=om, p_in=p_in, phase_in=phase,
                                                  beta_legacy=b_m * 1e3, length_unit="km", return_length_unit="m")
    # (b) GENERAL_TAYLOR single run
    pm_gen = PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4)
    z_g, A_g = simulation.run_single_simulation(cfg_m, gamma=0.0115, alpha=1.0e-4, omega=om, p_in=p_in, phase_in=phase,
                                                dispersion=disp, phase_matching_cfg=pm_gen)
    # (c), (d) the 4 x 9 grid row by row through the driver: symmetric (default) and GENERAL_TAYLOR
    rows_sym = pool.map(_g13_row, [(float(l2), lam3, dt, "sym") for l2 in lam2])
    rows_gen = pool.map(_g13_row, [(float(l2), lam3, dt, "gen") for l2 in lam2])
    _save("G13", omega=om, p_in=p_in, phase_in=phase, beta_legacy_m=b_m, disp=np.array(dt),
          legacy_m_z=z_m, legacy_m_A=A_m, legacy_km_z=z_km, legacy_km_A=A_km, gen_z=z_g, gen_A=A_g,
          lambda2=lam2, lambda3=lam3, grid_p_in=np.array([0.4, 0.3, 1e-6, 1e-6]),
          grid_gain_sym=np.array([r[0] for r in rows_sym]), grid_dbeta_sym=np.array([r[1] for r in rows_sym]),
          grid_gain_gen=np.array([r[0] for r in rows_gen]), grid_dbeta_gen=np.array([r[1] for r in rows_gen]))


def gen_api_signatures():
    """The call surface of the hot-path modules as data: for every public function its parameter names, kinds and
    defaults (as repr), for every public dataclass its field names.  Pins the drop-in boundary."""
    import dataclasses
    import importlib
    import inspect
    import json
    out = {}
    for modname in ("config", "integrators", "simulation", "scan_mismtach", "yaman_model", "parameters", "frequency_plan",
                    "dispersion", "phase_matching", "io_fwm"):
        m = importlib.import_module(modname)
        entry = {}
        for k, v in vars(m).items():
            if k.startswith("_") or not callable(v) or getattr(v, "__module__", "") != m.__name__:
                continue
            if inspect.isclass(v):
                entry[k] = {"kind": "class", "fields": [f.name for f in dataclasses.fields(v)] if dataclasses.is_dataclass(v) else None}
            else:
                entry[k] = {"kind": "function",
                            "params": [[n, q.kind.name, repr(q.default) if q.default is not inspect._empty else "<required>"]
                                       for n, q in inspect.signature(v).parameters.items()]}
        out[modname] = entry
    path = os.path.join(HERE, "api_signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"  wrote api_signatures.json ({os.path.getsize(path)} B)", flush=True)


def gen_error_contract():
    """What the reference raises (type + message) for the invalid calls of tests/golden/error_cases.py."""
    import importlib
    import json
    sys.path.insert(0, HERE)
    import error_cases
    out = error_cases.evaluate(importlib.import_module)
    path = os.path.join(HERE, "error_contract.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"  wrote error_contract.json ({len(out)} cases, {os.path.getsize(path)} B)", flush=True)


def gen_host_api_records():
    """What the reference returns (or raises) along the seeded random walk of tests/golden/host_api_cases.py."""
    import gzip
    import importlib
    import json
    sys.path.insert(0, HERE)
    import host_api_cases
    out = host_api_cases.evaluate(importlib.import_module)
    path = os.path.join(HERE, "host_api_records.json.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        json.dump(out, f, sort_keys=True)
    print(f"  wrote host_api_records.json.gz ({len(out)} records, {os.path.getsize(path)} B)", flush=True)


def _g15_run(args):
    import config, simulation
    from phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    (L, n, se, gamma, alpha, db, p_in, phase) = args
    cfg = config.custom_simulation_config(z_max=L, dz=L / n, save_every=se)
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=db)
    z, A = simulation.run_single_simulation(cfg, gamma=gamma, alpha=alpha, omega=np.full(4, 1.2e15), p_in=p_in, phase_in=phase,
                                            phase_matching_cfg=pm)
    return z, A


def gen_g15(pool):
    """SURVEY 8(d)'s robustness draw run through the REFERENCE: 32 single runs with dbeta, gamma, alpha, all four powers and
    phases, fibre length, step count and save stride drawn from default_rng(2026)."""
    rng = np.random.default_rng(2026)
    N = 32
    L = rng.uniform(50.0, 400.0, N)
    n = rng.integers(500, 3000, N)
    se = rng.choice([1, 3, 10, 32, 64, 100], N)
    gamma = rng.uniform(5e-3, 2e-2, N)
    alpha = rng.uniform(0.0, 3e-4, N)
    alpha[::5] = 0.0
    db = rng.uniform(-0.1, 0.1, N)
    P = np.stack([rng.uniform(0.05, 1, N), rng.uniform(0.05, 1, N), 10 ** rng.uniform(-7, -3, N), 10 ** rng.uniform(-7, -3, N)], 1)
    ph = rng.uniform(-np.pi, np.pi, (N, 4))
    ph[::4] = 0.0                                     # the all-zero-phase shortcut of make_initial_amplitudes
    res = pool.map(_g15_run, [(float(L[i]), int(n[i]), int(se[i]), float(gamma[i]), float(alpha[i]), float(db[i]), P[i], ph[i])
                              for i in range(N)])
    # n actually used upstream is int(round(L / dz)) with dz = L / n: record it from the saved grid
    a_end = np.array([r[1][-1] for r in res])
    p_max = np.array([np.max(np.abs(r[1][:, 2]) ** 2) for r in res])
    n_rows = np.array([r[1].shape[0] for r in res])
    z_last = np.array([r[0][-1] for r in res])
    full = {f"A_full_{i}": res[i][1] for i in (0, 7, 19, 31)}
    _save("G15", L=L, n=n, save_every=se, gamma=gamma, alpha=alpha, dbeta=db, p_in=P, phase_in=ph, A_end=a_end, p_max=p_max,
          n_rows=n_rows, z_last=z_last, **full)


def gen_g12():
    """Files written by the reference's io_fwm.save_run_bundle (tiny: 6 rows) -- pins the on-disk format."""
    import io_fwm
    z = np.linspace(0.0, 1.0, 6)
    A = (np.arange(24).reshape(6, 4) + 1j * np.arange(24).reshape(6, 4)[::-1]) * 0.1
    out = os.path.join(HERE, "G12_io_ref")
    io_fwm.save_run_bundle(out, "run", z, A, metadata={"gamma": 0.0115, "note": "written by the reference",
                                                        "timestamp_utc": "2026-01-01T00:00:00Z"}, overwrite=True)
    print("  wrote G12_io_ref/run.{npz,csv,json}", flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--procs", type=int, default=8)
    args = ap.parse_args()
    only = {s.strip().upper() for s in args.only.split(",") if s.strip()}
    want = lambda g: not only or g in only  # noqa: E731
    t0 = time.perf_counter()
    with Pool(args.procs) as pool:
        for gid, fn, needs_pool in [("G1", gen_g1, False), ("G4", gen_g4, False), ("G5", gen_g5, False),
                                    ("G6", gen_g6, False), ("G7", gen_g7, False), ("G9", gen_g9, False),
                                    ("G10", gen_g10, False), ("G12", gen_g12, False), ("API", gen_api_signatures, False), ("ERR", gen_error_contract, False), ("HOST", gen_host_api_records, False), ("G13", gen_g13, True), ("G15", gen_g15, True), ("G2", gen_g2, True), ("G3", gen_g3, True),
                                    ("G11", gen_g11, True), ("G8", gen_g8, True)]:
            if want(gid):
                print(f"{gid} ... ({time.perf_counter() - t0:.0f}s)", flush=True)
                fn(pool) if needs_pool else fn()
    print(f"done in {time.perf_counter() - t0:.0f}s")


if __name__ == "__main__":
    main()
