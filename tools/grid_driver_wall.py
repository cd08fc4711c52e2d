#!/usr/bin/env python3
"""Developer probe: wall time of scan_gain_grid on BASELINE config 3's 1024 x 1024 grid (host vs device dbeta producer)
against the kernel time inside it -- how much of the user-visible time is host work around the launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psa_amd import config, dispersion, frequency_plan, scan_mismtach
lam2 = np.linspace(1552e-9, 1562e-9, 1024)
lam3 = np.linspace(1540e-9, 1565e-9, 1024)
om = frequency_plan.plan_from_wavelengths(1550e-9, 1558e-9, 1540e-9)
sp = frequency_plan.infer_symmetry_from_omegas(*om)
d = dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.1, 0.02, 0, D_units="ps/nm/km",
                                          S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)
kw = dict(lambda_p1_m=1550e-9, lambda_p2_m=lam2, lambda_signal_m=lam3, gamma=0.0115, alpha=1.15e-4, p_in=[0.1, 0.1, 1e-7, 1e-7], dispersion=d)
scan_mismtach.scan_gain_grid(cfg=config.custom_simulation_config(z_max=1.0, dz=0.1), **kw)      # warm-up (context, module load)
for n_steps, dz in ((10_000, 0.1), (100_000, 0.01)):
    cfg = config.custom_simulation_config(z_max=1000.0, dz=dz)
    for producer in ("host", "device"):
        t = time.perf_counter()
        out = scan_mismtach.scan_gain_grid(cfg=cfg, dbeta_producer=producer, **kw)
        wall = time.perf_counter() - t
        k = out["result"].elapsed_ms
        print(f"{n_steps} steps, dbeta on {producer}: wall {wall * 1e3:.0f} ms, kernel {k:.0f} ms, host side {wall * 1e3 - k:.0f} ms "
              f"({(wall * 1e3 - k) / (wall * 1e3) * 100:.0f} %)", flush=True)
