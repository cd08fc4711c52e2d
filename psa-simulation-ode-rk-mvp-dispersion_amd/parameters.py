"""Parameter carriers handed to the integrator / RHS.

Same class and field names as the reference's parameters.py (WavesParams :90, FiberParams :166, SimulationGrid :209,
PhaseMatchingParams :224, CacheParams :236, ModelParams :254, factories :271-293) so user code ports unchanged; the
implementation is a small declarative layer: each carrier lists ``_SPEC = {field: rule}`` and one shared
``__post_init__`` normalises and checks the fields.

On the GPU path only five numbers of a ``ModelParams`` reach the kernel:
``fiber.gamma_W_m``, ``fiber.alpha_1_m``, ``fiber.length_m``, ``grid.dz_m`` and ``cache.delta_beta_1_m``
(see ``yaman_model.extract_gamma_alpha_dbeta``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .dispersion import DispersionParams
from .frequency_plan import SymmetricPlan, plan_from_omegas, plan_from_symmetry, plan_from_wavelengths  # noqa: F401
from .phase_matching import PhaseMatchingConfig, PhaseMatchingMethod

WAVE_ORDER: Tuple[str, str, str, str] = ("pump1", "pump2", "signal", "idler")


# ---- field rules ----------------------------------------------------------------------------------------------
def _number(lo: Optional[float] = None, *, open_lo: bool = False) -> Callable:
    """finite real scalar, optionally bounded below (open or closed)"""
    def rule(name, value):
        try:
            v = float(value)
        except Exception as exc:
            raise TypeError(f"{name}: expected a real scalar, got {type(value).__name__}") from exc
        if not np.isfinite(v):
            raise ValueError(f"{name}: not finite ({v!r})")
        if lo is not None and (v <= lo if open_lo else v < lo):
            raise ValueError(f"{name}: must be {'>' if open_lo else '>='} {lo:g} ({v!r})")
        return v
    return rule


def _four(positive: bool) -> Callable:
    """finite float vector of exactly four entries (wave order), optionally all > 0"""
    def rule(name, value):
        vec = np.asarray(list(value), dtype=float)
        if vec.shape != (4,):
            raise ValueError(f"{name}: expected 4 entries in wave order, got shape {vec.shape}")
        if not np.isfinite(vec).all():
            raise ValueError(f"{name}: entries must be finite")
        if positive and (vec <= 0.0).any():
            raise ValueError(f"{name}: angular frequencies must be > 0 rad/s")
        return vec
    return rule


def _instance(cls, *, optional: bool = False) -> Callable:
    def rule(name, value):
        if value is None and optional:
            return None
        if not isinstance(value, cls):
            raise TypeError(f"{name}: expected {cls.__name__}{' or None' if optional else ''}")
        return value
    return rule


def _maybe(inner: Callable) -> Callable:
    return lambda name, value: None if value is None else inner(name, value)


class _Carrier:
    """Shared normalise-and-check hook for the (frozen) dataclasses below."""
    _SPEC: Dict[str, Callable] = {}

    def __post_init__(self) -> None:
        for field, rule in self._SPEC.items():
            object.__setattr__(self, field, rule(field, getattr(self, field)))
        self._cross_check()

    def _cross_check(self) -> None:
        pass


# ---- carriers ----------------------------------------------------------------------------------------------------
@dataclass(slots=True)
class CacheParams:
    """Mutable on purpose: filled once per run with the mismatch that exp(+-i*dbeta*z) uses."""
    delta_beta_1_m: Optional[float] = None
    symmetric: Optional[SymmetricPlan] = None

    def set_phase_mismatch(self, delta_beta_1_m: float, symmetric: Optional[SymmetricPlan] = None) -> None:
        self.delta_beta_1_m = _number()("delta_beta_1_m", delta_beta_1_m)
        self.symmetric = symmetric


@dataclass(frozen=True)
class SimulationGrid(_Carrier):
    dz_m: float
    z0_m: float = 0.0
    _SPEC = {"dz_m": _number(0.0, open_lo=True), "z0_m": _number()}


@dataclass(frozen=True)
class PhaseMatchingParams(_Carrier):
    config: PhaseMatchingConfig
    _SPEC = {"config": _instance(PhaseMatchingConfig)}


@dataclass(frozen=True)
class FiberParams(_Carrier):
    length_m: float
    gamma_W_m: float
    alpha_1_m: float = 0.0                           # POWER attenuation; the field decays with alpha/2
    dispersion: Optional[DispersionParams] = None
    beta_legacy_1_m: Optional[np.ndarray] = None     # legacy beta(w_j); fallback dbeta = (b3 + b4) - (b1 + b2)
    _SPEC = {"length_m": _number(0.0, open_lo=True), "gamma_W_m": _number(), "alpha_1_m": _number(0.0),
             "dispersion": _instance(DispersionParams, optional=True), "beta_legacy_1_m": _maybe(_four(False))}


@dataclass(frozen=True)
class WavesParams(_Carrier):
    omega: np.ndarray                                # [w1, w2, w3, w4] rad/s
    symmetric: Optional[SymmetricPlan] = None        # optional (omega_c, omega_d, Omega) consistent with omega
    _SPEC = {"omega": _four(True), "symmetric": _instance(SymmetricPlan, optional=True)}

    def _cross_check(self) -> None:
        if self.symmetric is not None:
            rebuilt = self.symmetric.omegas()
            if not np.allclose(self.omega, rebuilt, rtol=1e-12, atol=0.0):
                raise ValueError(f"symmetric plan does not reproduce omega: {self.omega} vs {rebuilt}")

    omega1 = property(lambda self: float(self.omega[0]))
    omega2 = property(lambda self: float(self.omega[1]))
    omega3 = property(lambda self: float(self.omega[2]))
    omega4 = property(lambda self: float(self.omega[3]))

    @classmethod
    def from_symmetry(cls, omega_c: float, omega_d: float, Omega: float) -> "WavesParams":
        plan = SymmetricPlan(omega_c=omega_c, omega_d=omega_d, Omega=Omega)
        return cls(omega=plan.omegas(), symmetric=plan)

    @classmethod
    def from_omegas(cls, omega1, omega2, omega3, omega4=None) -> "WavesParams":
        return cls(omega=plan_from_omegas(omega1, omega2, omega3, omega4))

    @classmethod
    def from_wavelengths(cls, lambda1_m, lambda2_m, lambda3_m, lambda4_m=None) -> "WavesParams":
        return cls(omega=plan_from_wavelengths(lambda1_m, lambda2_m, lambda3_m, lambda4_m))


@dataclass(frozen=True)
class ModelParams(_Carrier):
    waves: WavesParams
    fiber: FiberParams
    grid: SimulationGrid
    phase_matching: PhaseMatchingParams
    cache: CacheParams
    _SPEC = {"cache": _instance(CacheParams)}


def make_default_phase_matching_params(*, method: PhaseMatchingMethod = PhaseMatchingMethod.SYMMETRIC_EVEN
                                       ) -> PhaseMatchingParams:
    return PhaseMatchingParams(PhaseMatchingConfig(method=method, max_order=4, even_orders=(2, 4)))


def make_model_params(*, waves: WavesParams, fiber: FiberParams, grid: SimulationGrid,
                      phase_matching: Optional[PhaseMatchingParams] = None) -> ModelParams:
    """Bundle the carriers with an empty cache (dbeta is computed once at run start and stored there)."""
    return ModelParams(waves, fiber, grid, phase_matching or make_default_phase_matching_params(),
                       CacheParams(None, waves.symmetric))
