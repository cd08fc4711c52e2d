"""How dbeta is obtained for a frequency plan -- the strategy layer of the reference's phase_matching.py
(PhaseMatchingMethod :50-53, PhaseMatchingConfig :77-138, compute_phase_mismatch :150-215,
PhaseMismatchCalculator :218-243), plus ``compute_phase_mismatch_batch`` which serves a whole sweep
(dbeta[N] + validity mask) in one call and is what feeds the GPU kernel.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
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


@dataclass(frozen=True)
class PhaseMatchingResult:
    delta_beta: float
    symmetric: Optional[SymmetricPlan] = None


# ---- PhaseMatchingConfig: field rules as data ----------------------------------------------------------------------
# Each rule takes the raw field value and returns what is stored (or raises); cross-field constraints follow.  The messages
# are the reference's (tests/golden/error_contract.json pins type and text of what upstream raises, phase_matching.py:106-137).
def _finite_real(name):
    def rule(x):
        try:
            v = float(x)
        except Exception as e:
            raise TypeError(f"{name} must be a real scalar, got {type(x)!r}") from e
        if not np.isfinite(v):
            raise ValueError(f"{name} must be finite, got {v!r}")
        return v
    return rule


def _method_rule(m):
    if isinstance(m, PhaseMatchingMethod):
        return m
    try:
        return PhaseMatchingMethod(str(m))
    except Exception as e:
        raise ValueError(f"Invalid method {m!r}") from e


def _max_order_rule(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"max_order must be int >= 0, got {n!r}")
    return n


def _even_orders_rule(orders):
    seq = tuple(orders)
    if not seq:
        raise ValueError("even_orders must not be empty (e.g., (2,4))")
    for n in seq:
        if not isinstance(n, int):
            raise TypeError("even_orders must contain ints")
        if n < 2 or n % 2:
            raise ValueError(f"even_orders must contain even ints >= 2, got {n!r}")
    return orders          # stored as given


_FIELD_RULES = (("method", _method_rule), ("max_order", _max_order_rule), ("even_orders", _even_orders_rule),
                ("atol", _finite_real("atol")), ("rtol", _finite_real("rtol")))
_PROVIDED_RULE = _finite_real("provided_delta_beta")


@dataclass(frozen=True)
class PhaseMatchingConfig:
    method: PhaseMatchingMethod = PhaseMatchingMethod.SYMMETRIC_EVEN
    max_order: int = 4                        # GENERAL_TAYLOR: highest Taylor order
    even_orders: Tuple[int, ...] = (2, 4)     # SYMMETRIC_EVEN: orders summed, in this order
    atol: float = 0.0                         # energy-conservation tolerances
    rtol: float = 1e-12
    provided_delta_beta: Optional[float] = None   # PROVIDED only

    def __post_init__(self) -> None:
        for name, rule in _FIELD_RULES:
            object.__setattr__(self, name, rule(getattr(self, name)))
        if min(self.atol, self.rtol) < 0.0:
            raise ValueError("atol and rtol must be >= 0")
        if self.method is PhaseMatchingMethod.PROVIDED:
            if self.provided_delta_beta is None:
                raise ValueError("provided_delta_beta must be set when method == 'provided'")
            object.__setattr__(self, "provided_delta_beta", _PROVIDED_RULE(self.provided_delta_beta))

    def scaled(self, length_scale: float) -> "PhaseMatchingConfig":
        """PROVIDED dbeta per (length unit / length_scale) -- simulation.py:153-175; other methods unchanged."""
        s = float(length_scale)
        if self.method != PhaseMatchingMethod.PROVIDED or s == 1.0:
            return self
        return replace(self, provided_delta_beta=float(self.provided_delta_beta) / s)


# ---- one strategy per method, scalar (one plan, raises) and batched (a sweep, masks) ---------------------------------
def _omega4(omegas, name="omegas") -> np.ndarray:
    arr = np.asarray(list(omegas), dtype=float)
    if arr.shape != (4,):
        raise ValueError(f"{name} must have shape (4,), got {arr.shape}")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must contain only finite values")
    if np.any(arr <= 0.0):
        raise ValueError(f"{name} must contain only positive angular frequencies (rad/s)")
    return arr


def _one_provided(om, disp, cfg, hint):
    return PhaseMatchingResult(float(cfg.provided_delta_beta), None)


def _one_taylor(om, disp, cfg, hint):
    db = delta_beta_from_omegas(om, disp, max_order=cfg.max_order, atol=cfg.atol, rtol=cfg.rtol)
    return PhaseMatchingResult(float(db), None)


def _one_symmetric(om, disp, cfg, hint):
    sp = hint if hint is not None else infer_symmetry_from_omegas(*(float(w) for w in om), atol=cfg.atol, rtol=cfg.rtol)
    return PhaseMatchingResult(float(delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, disp,
                                                          even_orders=cfg.even_orders)), sp)


def _many_provided(om, disp, cfg):
    return np.full(om.shape[0], float(cfg.provided_delta_beta)), True


def _many_taylor(om, disp, cfg):
    ok = _conserves(om[:, 0] + om[:, 1], om[:, 2] + om[:, 3], cfg.atol, cfg.rtol)
    return delta_beta_from_omegas_array(om, disp, max_order=cfg.max_order), ok


def _many_symmetric(om, disp, cfg):
    _, od, Om, ok = symmetry_arrays(om[:, 0], om[:, 1], om[:, 2], om[:, 3], atol=cfg.atol, rtol=cfg.rtol)
    return delta_beta_symmetric_array(od, Om, disp, even_orders=cfg.even_orders), ok


# method -> (scalar strategy, batched strategy, needs a dispersion model)
_STRATEGIES = {
    PhaseMatchingMethod.PROVIDED: (_one_provided, _many_provided, False),
    PhaseMatchingMethod.GENERAL_TAYLOR: (_one_taylor, _many_taylor, True),
    PhaseMatchingMethod.SYMMETRIC_EVEN: (_one_symmetric, _many_symmetric, True),
}


def _strategy(cfg, disp):
    try:
        one, many, needs_disp = _STRATEGIES[cfg.method]
    except KeyError:
        raise ValueError(f"Unsupported phase-matching method: {cfg.method!r}") from None
    if needs_disp and disp is None:
        raise ValueError("disp must be provided unless method == 'provided'")
    return one, many


def compute_phase_mismatch(omegas: Sequence[float], disp: Optional[DispersionParams], cfg: PhaseMatchingConfig, *,
                           symmetric_hint: Optional[SymmetricPlan] = None) -> PhaseMatchingResult:
    """dbeta for one plan [w1, w2, w3, w4]; raises like the reference (phase_matching.py:177-215)."""
    om = _omega4(omegas)
    return _strategy(cfg, disp)[0](om, disp, cfg, symmetric_hint)


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
    many = _strategy(cfg, disp)[1]
    with np.errstate(all="ignore"):
        db, ok_method = many(om, disp, cfg)
        ok = np.all(np.isfinite(om), axis=1) & np.all(om > 0.0, axis=1) & ok_method & np.isfinite(db)
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
