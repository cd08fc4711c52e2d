#!/usr/bin/env python3
"""Developer probe: device dbeta producer vs the host NumPy producer on a large random grid (how often are they bit-equal,
how far apart at worst).  The host array path itself is within an ulp of the reference's scalar path (array pow)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psa_amd._native as nat
from psa_amd import dispersion, frequency_plan, phase_matching
from psa_amd.phase_matching import PhaseMatchingConfig

rng = np.random.default_rng(2)
lam2 = np.sort(rng.uniform(1545e-9, 1570e-9, 1000))
lam3 = np.sort(rng.uniform(1500e-9, 1620e-9, 1000))
d = dispersion.DispersionParams(omega_ref=frequency_plan.omega_from_lambda(1552e-9), beta2=-2.3e-28, beta3=4.1e-41, beta4=-3.0e-55,
                                extra={6: 1.0e-84})
L2, L3 = np.meshgrid(lam2, lam3, indexing="ij")
om, ok = frequency_plan.plan_from_wavelengths_batch(1550e-9, L2.ravel(), L3.ravel())
for name, cfg in (("symmetric (2,4)", PhaseMatchingConfig()), ("symmetric (2,)", PhaseMatchingConfig(even_orders=(2,))),
                  ("symmetric (2,4,6)", PhaseMatchingConfig(even_orders=(2, 4, 6))),
                  ("taylor 4", PhaseMatchingConfig(method="general_taylor", max_order=4)),
                  ("taylor 2", PhaseMatchingConfig(method="general_taylor", max_order=2))):
    ref, ok2 = phase_matching.compute_phase_mismatch_batch(om, d, cfg)
    dev, okd = nat.dbeta_grid_host(nat.dbeta_model(d, cfg), 1550e-9, lam2, lam3)
    good = ok & ok2
    assert np.array_equal(okd, good)
    u = np.abs(dev[good] - ref[good]) / np.spacing(np.abs(ref[good]))
    rel = np.abs(dev[good] - ref[good]) / np.abs(ref[good])
    print(f"{name:18s}: {good.sum()} valid points, bit-equal {np.mean(u == 0) * 100:.3f} %, <= 1 ulp {np.mean(u <= 1) * 100:.4f} %, "
          f"max {u.max():.1f} ulp, max rel {rel.max():.2e}", flush=True)

# ---- validity: wavelengths far outside the band, so that a large fraction of the plans is impossible (omega_4 <= 0,
# |omega_d| >= omega_c, energy conservation, non-finite dbeta); the device mask must equal the host mask exactly
lam2w = np.concatenate([np.sort(rng.uniform(0.4e-6, 4e-6, 996)), [0.0, -1e-6, np.nan, np.inf]])
lam3w = np.concatenate([np.sort(rng.uniform(0.3e-6, 6e-6, 997)), [0.0, -2e-6, np.nan]])
L2, L3 = np.meshgrid(lam2w, lam3w, indexing="ij")
with np.errstate(all="ignore"):
    om, ok = frequency_plan.plan_from_wavelengths_batch(1550e-9, L2.ravel(), L3.ravel())
    for name, cfg in (("symmetric (2,4)", PhaseMatchingConfig()), ("taylor 4", PhaseMatchingConfig(method="general_taylor", max_order=4))):
        ref, ok2 = phase_matching.compute_phase_mismatch_batch(om, d, cfg)
        dev, okd = nat.dbeta_grid_host(nat.dbeta_model(d, cfg), 1550e-9, lam2w, lam3w)
        good = ok & ok2
        same = np.array_equal(okd, good)
        nan_ok = np.array_equal(np.isnan(dev), ~good)
        print(f"validity {name:16s}: {good.sum()} valid of {good.size}, masks equal: {same}, NaN exactly where invalid: {nan_ok}", flush=True)
        assert same and nan_ok
