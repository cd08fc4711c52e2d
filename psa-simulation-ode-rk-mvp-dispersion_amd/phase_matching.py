"""How dbeta is obtained for a frequency plan -- the strategy layer of the reference's phase_matching.py
(PhaseMatchingMethod :50-53, PhaseMatchingConfig :77-138, compute_phase_mismatch :150-215,
PhaseMismatchCalculator :218-243), plus ``compute_phase_mismatch_batch`` which serves a whole sweep
(dbeta[N] + validity mask) in one call and is what feeds the GPU kernel.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Optional, Sequence, Tuple

import numpy as np

from .dispersion import (DispersionParams, delta_beta_from_omegas, delta_beta_from_omegas_array,
                         delta_beta_symmetric, delta_beta_symmetric_array)
from .frequency_plan import SymmetricPlan, infer_symmetry_from_omegas, symmetry_arrays, _conserves


class PhaseMatchingMethod(str, Enum):
    GENERAL_TAYLOR = "general_taylor"   # beta(w_j) by Taylor series, assembled from the four omegas
    SYMMETRIC_EVEN = "symmetric_even"   # closed form in (omega_d, Omega): even orders only (default)
    PROVIDED = "provided"               # a constant supplied by the caller


def _nonneg_real(x, name):
    try:
        v = float(x)
    except Exception as e:
        raise TypeError(f"{name} must be a real scalar, got {type(x)!r}") from e
    if not np.isfinite(v):
        raise ValueError(f"{name} must be finite, got {v!r}")
    return v


@dataclass(frozen=True)
class PhaseMatchingConfig:
    method: PhaseMatchingMethod = PhaseMatchingMethod.SYMMETRIC_EVEN
    max_order: int = 4                        # GENERAL_TAYLOR: highest Taylor order
    even_orders: Tuple[int, ...] = (2, 4)     # SYMMETRIC_EVEN: orders summed, in this order
    atol: float = 0.0                         # energy-conservation tolerances
    rtol: float = 1e-12
    provided_delta_beta: Optional[float] = None   # PROVIDED only

    def __post_init__(self) -> None:
        if not isinstance(self.method, PhaseMatchingMethod):
            try:
                object.__setattr__(self, "method", PhaseMatchingMethod(str(self.method)))
            except Exception as e:
                raise ValueError(f"Invalid method {self.method!r}") from e
        if not isinstance(self.max_order, int) or self.max_order < 0:
            raise ValueError(f"max_order must be int >= 0, got {self.max_order!r}")
        orders = tuple(self.even_orders)
        if not orders:
            raise ValueError("even_orders must not be empty (e.g., (2,4))")
        for n in orders:
            if not isinstance(n, int):
                raise TypeError("even_orders must contain ints")
            if n < 2 or n % 2:
                raise ValueError(f"even_orders must contain even ints >= 2, got {n!r}")
        a, r = _nonneg_real(self.atol, "atol"), _nonneg_real(self.rtol, "rtol")
        if a < 0.0 or r < 0.0:
            raise ValueError("atol and rtol must be >= 0")
        object.__setattr__(self, "atol", a)
        object.__setattr__(self, "rtol", r)
        if self.method == PhaseMatchingMethod.PROVIDED:
            if self.provided_delta_beta is None:
                raise ValueError("provided_delta_beta must be set when method == 'provided'")
            object.__setattr__(self, "provided_delta_beta",
                               _nonneg_real(self.provided_delta_beta, "provided_delta_beta"))

    def scaled(self, length_scale: float) -> "PhaseMatchingConfig":
        """PROVIDED dbeta per (length unit / length_scale) -- simulation.py:153-175; other methods unchanged."""
        s = float(length_scale)
        if self.method != PhaseMatchingMethod.PROVIDED or s == 1.0:
            return self
        return PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, max_order=self.max_order,
                                   even_orders=self.even_orders, atol=self.atol, rtol=self.rtol,
                                   provided_delta_beta=float(self.provided_delta_beta) / s)


@dataclass(frozen=True)
class PhaseMatchingResult:
    delta_beta: float
    symmetric: Optional[SymmetricPlan] = None


def _omega4(omegas, name="omegas") -> np.ndarray:
    arr = np.asarray(list(omegas), dtype=float)
    if arr.shape != (4,):
        raise ValueError(f"{name} must have shape (4,), got {arr.shape}")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must contain only finite values")
    if np.any(arr <= 0.0):
        raise ValueError(f"{name} must contain only positive angular frequencies (rad/s)")
    return arr


def compute_phase_mismatch(omegas: Sequence[float], disp: Optional[DispersionParams], cfg: PhaseMatchingConfig, *,
                           symmetric_hint: Optional[SymmetricPlan] = None) -> PhaseMatchingResult:
    """dbeta for one plan [w1, w2, w3, w4]; raises like the reference (phase_matching.py:177-215)."""
    om = _omega4(omegas)
    if cfg.method == PhaseMatchingMethod.PROVIDED:
        return PhaseMatchingResult(float(cfg.provided_delta_beta), None)
    if disp is None:
        raise ValueError("disp must be provided unless method == 'provided'")
    if cfg.method == PhaseMatchingMethod.GENERAL_TAYLOR:
        return PhaseMatchingResult(float(delta_beta_from_omegas(om, disp, max_order=cfg.max_order, atol=cfg.atol,
                                                                rtol=cfg.rtol)), None)
    if cfg.method == PhaseMatchingMethod.SYMMETRIC_EVEN:
        sp = symmetric_hint
        if sp is None:
            sp = infer_symmetry_from_omegas(float(om[0]), float(om[1]), float(om[2]), float(om[3]),
                                            atol=cfg.atol, rtol=cfg.rtol)
        return PhaseMatchingResult(float(delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, disp,
                                                              even_orders=cfg.even_orders)), sp)
    raise ValueError(f"Unsupported phase-matching method: {cfg.method!r}")


def compute_phase_mismatch_batch(omega: np.ndarray, disp: Optional[DispersionParams], cfg: PhaseMatchingConfig
                                 ) -> Tuple[np.ndarray, np.ndarray]:
    """All sweep points at once: omega[N,4] -> (dbeta[N], valid[N]).

    ``valid`` is False exactly where ``compute_phase_mismatch`` would raise for that row (bad omegas, energy
    conservation, inconsistent symmetric plan) or where dbeta comes out non-finite (the cache setter
    rejects it, parameters.py:248); dbeta is NaN there.  A missing ``disp`` is a caller error and raises.
    """
    om = np.asarray(omega, dtype=float)
    if om.ndim != 2 or om.shape[1] != 4:
        raise ValueError(f"omega must have shape (N, 4), got {om.shape}")
    with np.errstate(all="ignore"):
        ok = np.all(np.isfinite(om), axis=1) & np.all(om > 0.0, axis=1)
        if cfg.method == PhaseMatchingMethod.PROVIDED:
            db = np.full(om.shape[0], float(cfg.provided_delta_beta))
        else:
            if disp is None:
                raise ValueError("disp must be provided unless method == 'provided'")
            if cfg.method == PhaseMatchingMethod.GENERAL_TAYLOR:
                ok &= _conserves(om[:, 0] + om[:, 1], om[:, 2] + om[:, 3], cfg.atol, cfg.rtol)
                db = delta_beta_from_omegas_array(om, disp, max_order=cfg.max_order)
            else:
                _, od, Om, sym_ok = symmetry_arrays(om[:, 0], om[:, 1], om[:, 2], om[:, 3], atol=cfg.atol,
                                                    rtol=cfg.rtol)
                ok &= sym_ok
                db = delta_beta_symmetric_array(od, Om, disp, even_orders=cfg.even_orders)
        ok &= np.isfinite(db)
        db = np.where(ok, db, np.nan)
    return db, ok


@dataclass(frozen=True)
class PhaseMismatchCalculator:
    """Callable with a fixed (disp, cfg); ``calc(omegas)`` == ``compute_phase_mismatch(omegas, disp, cfg)``."""
    disp: Optional[DispersionParams]
    cfg: PhaseMatchingConfig

    def __call__(self, omegas: Sequence[float], *, symmetric_hint: Optional[SymmetricPlan] = None
                 ) -> PhaseMatchingResult:
        return compute_phase_mismatch(omegas, self.disp, self.cfg, symmetric_hint=symmetric_hint)

    def batch(self, omega: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        return compute_phase_mismatch_batch(omega, self.disp, self.cfg)
