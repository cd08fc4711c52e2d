"""Single-point runner with the reference's call surface (simulation.py:220-364).

``run_single_simulation(cfg, *, gamma, alpha, omega, p_in, ...) -> (z_out, A[n_saved, 4])``: validates,
converts ``length_unit`` quantities to metres (gamma, alpha, dbeta, beta_n divided by the scale; lengths
multiplied), builds A0 = sqrt(P)*exp(i*phi), computes dbeta ONCE on the host, then hands the whole
propagation to the HIP kernel through ``integrators.integrate_interval(rhs_yaman_simplified, ...)``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import constants
from .config import (SimulationConfig, custom_simulation_config, default_simulation_config,  # noqa: F401
                     validate_config)
from .dispersion import DispersionParams
from .integrators import integrate_interval
from .parameters import FiberParams, PhaseMatchingParams, SimulationGrid, WavesParams, make_model_params
from .phase_matching import PhaseMatchingConfig, PhaseMatchingMethod, PhaseMatchingResult, compute_phase_mismatch  # noqa: F401
from .sweep import initial_amplitudes
from .yaman_model import rhs_yaman_simplified

_UNITS = {"m": 1.0, "km": 1000.0}


def _length_scale_to_m(length_unit: str) -> float:
    try:
        return _UNITS[str(length_unit).strip().lower()]
    except KeyError:
        raise ValueError(f"Unsupported length_unit={length_unit!r}. Use 'm' or 'km'.") from None


def _vec4(x, name: str, *, what: str, lower: Optional[float] = None, strict: bool = False) -> np.ndarray:
    arr = np.asarray(list(x), dtype=float)
    if arr.shape != (4,):
        raise ValueError(f"{name} must have shape (4,), got {arr.shape}")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must be finite")
    if lower is not None and np.any(arr <= lower if strict else arr < lower):
        raise ValueError(f"{name} must be {what}")
    return arr


def _to_omega_array(omega) -> np.ndarray:
    return _vec4(omega, "omega", what="positive (rad/s)", lower=0.0, strict=True)


def _to_power_array(p_in) -> np.ndarray:
    return _vec4(p_in, "p_in", what="non-negative (W)", lower=0.0)


def _to_phase_array(phase_in) -> np.ndarray:
    return np.zeros(4) if phase_in is None else _vec4(phase_in, "phase_in", what="")


def make_initial_amplitudes(p_in: Sequence[float], phase_in: Optional[Sequence[float]] = None) -> np.ndarray:
    """|A_j|^2 = P_j, A_j = sqrt(P_j)*exp(i*phi_j) -> complex128 (4,)  (simulation.py:103-123)."""
    return initial_amplitudes(_to_power_array(p_in), _to_phase_array(phase_in))


def _default_phase_matching_cfg(*, dispersion, beta_legacy) -> PhaseMatchingConfig:
    """dispersion -> SYMMETRIC_EVEN (2,4); only legacy betas -> PROVIDED b3+b4-b1-b2; neither -> ValueError."""
    if dispersion is not None:
        return PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, max_order=4, even_orders=(2, 4),
                                   atol=0.0, rtol=1e-12)
    if beta_legacy is not None:
        b = np.asarray(beta_legacy, dtype=float)
        if b.shape != (4,):
            raise ValueError("beta_legacy must have shape (4,)")
        return PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, max_order=0, even_orders=(2,), atol=0.0,
                                   rtol=1e-12, provided_delta_beta=float((b[2] + b[3]) - (b[0] + b[1])))
    raise ValueError("Provide either dispersion or beta_legacy (or an explicit phase_matching_cfg).")


def _prepare(cfg, *, gamma, alpha, dispersion, phase_matching_cfg, beta_legacy, length_unit):
    """Everything of run_single_simulation that does not depend on the frequency plan (shared with the sweep
    drivers, which run it once instead of once per point).  Returns a dict of per-metre quantities."""
    validate_config(cfg)
    scale = _length_scale_to_m(length_unit)
    legacy_m = None
    if beta_legacy is not None:
        b = np.asarray(list(beta_legacy), dtype=float)
        if b.shape != (4,):
            raise ValueError(f"beta_legacy must have shape (4,), got {b.shape}")
        if not np.all(np.isfinite(b)):
            raise ValueError("beta_legacy must be finite")
        legacy_m = b / scale
    disp_m = None
    if dispersion is not None:
        if not isinstance(dispersion, DispersionParams):
            raise TypeError("dispersion must be DispersionParams or None")
        disp_m = dispersion.scaled(scale)
    pm_cfg = phase_matching_cfg if phase_matching_cfg is not None else \
        _default_phase_matching_cfg(dispersion=disp_m, beta_legacy=legacy_m)
    if not isinstance(pm_cfg, PhaseMatchingConfig):
        raise TypeError("phase_matching_cfg must be PhaseMatchingConfig or None")
    fiber = FiberParams(length_m=float(cfg.z_max) * scale, gamma_W_m=float(gamma) / scale,
                        alpha_1_m=float(alpha) / scale, dispersion=disp_m, beta_legacy_1_m=legacy_m)
    grid = SimulationGrid(dz_m=float(cfg.dz) * scale, z0_m=0.0)
    return dict(scale=scale, fiber=fiber, grid=grid, pm=PhaseMatchingParams(config=pm_cfg.scaled(scale)))


def run_single_simulation(cfg: SimulationConfig, *, gamma: float, alpha: float, omega: Sequence[float],
                          p_in: Sequence[float], phase_in: Optional[Sequence[float]] = None,
                          dispersion: Optional[DispersionParams] = None,
                          phase_matching_cfg: Optional[PhaseMatchingConfig] = None,
                          beta_legacy: Optional[Sequence[float]] = None, length_unit: str = "m",
                          return_length_unit: Optional[str] = None) -> tuple[np.ndarray, np.ndarray]:
    """One propagation on the GPU -> (z_out in ``return_length_unit``, A complex128 (n_saved, 4))."""
    validate_config(cfg)
    _length_scale_to_m(length_unit)
    om = _to_omega_array(omega)
    A0 = make_initial_amplitudes(_to_power_array(p_in), phase_in)
    pre = _prepare(cfg, gamma=gamma, alpha=alpha, dispersion=dispersion, phase_matching_cfg=phase_matching_cfg,
                   beta_legacy=beta_legacy, length_unit=length_unit)
    params = make_model_params(waves=WavesParams(omega=om, symmetric=None), fiber=pre["fiber"], grid=pre["grid"],
                               phase_matching=pre["pm"])
    res = compute_phase_mismatch(params.waves.omega, params.fiber.dispersion, params.phase_matching.config,
                                 symmetric_hint=params.waves.symmetric)
    params.cache.set_phase_mismatch(res.delta_beta, symmetric=res.symmetric)
    z_m, A = integrate_interval(rhs_yaman_simplified, params.fiber.length_m, params.grid.dz_m, A0, params,
                                save_every=cfg.save_every, check_nan=cfg.check_nan)
    out_unit = length_unit if return_length_unit is None else return_length_unit
    return z_m / _length_scale_to_m(out_unit), A


# ---- the reference's two ready-made scenarios (simulation.py:371-447), km-unit path -----------------------
def _omega_1550x4() -> np.ndarray:
    return np.full(4, 2.0 * np.pi * constants.c / 1.55e-6)


def example_zero_signal() -> tuple[np.ndarray, np.ndarray]:
    """Two 0.5 W pumps, no signal/idler, dbeta = 0, gamma = 1.3 /(W km), 0.5 km in 1e-3 km steps."""
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=0.0)
    return run_single_simulation(default_simulation_config(), gamma=1.3, alpha=0.0, omega=_omega_1550x4(),
                                 p_in=np.array([0.5, 0.5, 0.0, 0.0]), phase_matching_cfg=pm, length_unit="km",
                                 return_length_unit="km")


def custom_seeded_signal() -> tuple[np.ndarray, np.ndarray]:
    """0.1 W pumps, 1e-4 / 1e-6 W seeds, dbeta = 0, gamma = 10 /(W km), 0.5 km in 1e-4 km steps."""
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=0.0)
    return run_single_simulation(custom_simulation_config(z_max=0.5, dz=1e-4), gamma=10.0, alpha=0.0,
                                 omega=_omega_1550x4(), p_in=np.array([1e-1, 1e-1, 1e-4, 1e-6]), phase_in=np.zeros(4),
                                 phase_matching_cfg=pm, length_unit="km", return_length_unit="km")
