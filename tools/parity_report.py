#!/usr/bin/env python3
"""Measured parity of the HIP path (through the C-ABI) against the reference's golden vectors and the oracle.
Prints the numbers quoted in DESIGN.md section 4; the pass/fail versions of the same checks live in tests/."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import psa_amd._native as nat
from psa_amd import config, dispersion, frequency_plan, scan_mismtach, simulation
from psa_amd.phase_matching import PhaseMatchingConfig

G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
a0of = lambda p: np.sqrt(np.asarray(p, float)).astype(complex)
print(nat.version(), "| devices:", nat.device_count())

g = G("G1")
om = frequency_plan.plan_from_wavelengths(*g["lam"]); sp = frequency_plan.infer_symmetry_from_omegas(*om)
d = dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.02, 0.02, 0, D_units="ps/nm/km",
                                          S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)
z, A = simulation.run_single_simulation(config.custom_simulation_config(z_max=1000.0, dz=0.1), gamma=float(g["gamma"]),
                                        alpha=float(g["alpha"]), omega=om, p_in=g["p_in"], dispersion=d,
                                        phase_matching_cfg=PhaseMatchingConfig())
gain = 10 * np.log10(abs(A[-1, 2]) ** 2 / g["p_in"][2])
print(f"G1  run_single_simulation, 10 000 steps, 1001 rows: max rel err over all rows {rel(A, g['A']):.2e}; "
      f"gain {gain:.12f} dB vs reference {float(g['gain_db']):.12f} dB (diff {abs(gain - float(g['gain_db'])):.1e})")

for name, drv in (("G2", "gain+dbeta driver"), ("G3", "gain driver")):
    g = G(name)
    dd = dispersion.DispersionParams(omega_ref=float(g["omega_ref"]), beta2=float(g["beta2"]), beta3=float(g["beta3"]), beta4=float(g["beta4"]))
    kw = dict(cfg=config.custom_simulation_config(z_max=500.0, dz=0.2), lambda_p1_m=float(g["lambda_p1"]), lambda_p2_m=float(g["lambda_p2"]),
              lambda_signal_m=g["lambda3"], gamma=float(g["gamma"]), alpha=float(g["alpha"]), p_in=g["p_in"], dispersion=dd, show=False)
    if name == "G2":
        x, gn, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(**kw)
        extra = f"; dbeta bit-exact: {np.array_equal(db, g['dbeta'])}"
    else:
        x, gn = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**kw); extra = ""
    print(f"{name}  {drv}, {gn.size} points x 2500 steps: max |gain - ref| = {np.max(np.abs(gn - g['gain_db'])):.2e} dB "
          f"(peak {g['gain_db'].max():.4f} dB){extra}")

g = G("G8")
for key, n, ai, dbk in (("n1e4_a0", 10_000, 0, "dbeta257"), ("n1e4_a1", 10_000, 1, "dbeta257"), ("n1e5_a1", 100_000, 1, "dbeta33")):
    r = nat.sweep_host(g[dbk], n_steps=n, z_max=1000.0, save_every=10, gamma=float(g["gamma"]), alpha=float(g["alphas"][ai]), a0=a0of(g["p_in"]))
    print(f"G8  {g[dbk].size} points x {n} steps alpha={g['alphas'][ai]:g}: A_end rel err {rel(r['a_end'], g[key + '_A_end']):.2e}, "
          f"|A3|^2 end {rel(r['p_end'], g[key + '_p_end']):.2e}, max {rel(r['p_max'], g[key + '_p_max']):.2e}")

g = G("G9")
r = nat.sweep_host(np.full(g["gammas"].size, float(g["dbeta"])), n_steps=1000, z_max=100.0, save_every=10, gamma=g["gammas"], alpha=0.0,
                   a0=a0of(g["p_in"]), check_nan=True, exact_step=True)
print(f"G9  first non-finite step: reference {g['first_bad_step'].tolist()} | HIP {r['first_bad_step'].tolist()}")

g = G("G5")
out = nat.yaman_rhs_host(g["z"], g["a"], g["gamma"], g["alpha"], g["dbeta"])
print(f"G5  64 direct RHS evaluations: max err / max|rhs| per point {np.max(np.abs(out - g['rhs']).max(1) / np.abs(g['rhs']).max(1)):.2e}")

N, n = 65536, 100_000
db = np.linspace(-0.05, 0.05, N); a0 = a0of([0.5, 0.5, 1e-5, 1e-5])
r = nat.sweep_host(db, n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
pick = np.random.default_rng(1).choice(N, 64, replace=False)
ref = O.sweep(db[pick], z_max=1000.0, n=n, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
P = np.abs(r["a_end"]) ** 2
print(f"C2  65 536 points x 100 000 steps (bench workload), 64 sampled points vs oracle: A_end rel err {rel(r['a_end'][pick], ref['a_end']):.2e}; "
      f"power balance max rel dev {np.max(np.abs(P.sum(1) / (1.00002 * np.exp(-0.115)) - 1)):.2e}; kernel {r['elapsed_ms']:.1f} ms")
r32 = nat.sweep_host(db, n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0, dtype=np.float32)
print(f"C2 in float32 (packed, Kahan state) vs float64 kernel: A_end rel err {rel(r32['a_end'].astype(complex), r['a_end']):.2e}, "
      f"gain dB err {np.max(np.abs(10*np.log10(r32['p_max'].astype(float)/r['p_max']))):.1e}; kernel {r32['elapsed_ms']:.1f} ms")
