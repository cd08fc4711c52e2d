"""Parameter carriers handed to the integrator / RHS -- the shapes of the reference's parameters.py
(WavesParams :90-163, FiberParams :166-206, SimulationGrid :209-221, PhaseMatchingParams :224-233,
CacheParams :236-251, ModelParams :254-268, factories :271-293).

On the GPU path only five numbers of a ``ModelParams`` reach the kernel:
gamma = fiber.gamma_W_m, alpha = fiber.alpha_1_m, L = fiber.length_m, dz = grid.dz_m and
dbeta = cache.delta_beta_1_m (see yaman_model.extract_gamma_alpha_dbeta).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from .dispersion import DispersionParams
from .frequency_plan import SymmetricPlan, plan_from_omegas, plan_from_symmetry, plan_from_wavelengths  # noqa: F401
from .phase_matching import PhaseMatchingConfig, PhaseMatchingMethod

WAVE_ORDER: Tuple[str, str, str, str] = ("pump1", "pump2", "signal", "idler")


def _vec4(values: Sequence[float], name: str, positive: bool) -> np.ndarray:
    arr = np.asarray(list(values), dtype=float)
    if arr.shape != (4,):
        raise ValueError(f"{name} must have shape (4,), got {arr.shape}")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must contain finite values")
    if positive and np.any(arr <= 0.0):
        raise ValueError(f"{name} must contain positive angular frequencies (rad/s)")
    return arr


def _scalar(x, name: str, *, minimum: Optional[float] = None, strict: bool = False) -> float:
    try:
        v = float(x)
    except Exception as e:
        raise TypeError(f"{name} must be a real scalar, got {type(x)!r}") from e
    if not np.isfinite(v):
        raise ValueError(f"{name} must be finite, got {v!r}")
    if minimum is not None and (v <= minimum if strict else v < minimum):
        raise ValueError(f"{name} must be {'>' if strict else '>='} {minimum:g}, got {v!r}")
    return v


@dataclass(frozen=True, slots=True)
class WavesParams:
    omega: np.ndarray                          # [w1, w2, w3, w4] rad/s
    symmetric: Optional[SymmetricPlan] = None  # optional (omega_c, omega_d, Omega) consistent with omega

    def __post_init__(self) -> None:
        om = _vec4(self.omega, "omega", positive=True)
        object.__setattr__(self, "omega", om)
        if self.symmetric is not None:
            if not isinstance(self.symmetric, SymmetricPlan):
                raise TypeError("symmetric must be SymmetricPlan or None")
            regenerated = self.symmetric.omegas()
            if not np.allclose(om, regenerated, rtol=1e-12, atol=0.0):
                raise ValueError("Provided symmetric plan is inconsistent with omega. "
                                 f"omega={om}, omega(sym)={regenerated}")

    omega1 = property(lambda self: float(self.omega[0]))
    omega2 = property(lambda self: float(self.omega[1]))
    omega3 = property(lambda self: float(self.omega[2]))
    omega4 = property(lambda self: float(self.omega[3]))

    @classmethod
    def from_symmetry(cls, omega_c: float, omega_d: float, Omega: float) -> "WavesParams":
        sp = SymmetricPlan(omega_c=omega_c, omega_d=omega_d, Omega=Omega)
        return cls(omega=sp.omegas(), symmetric=sp)

    @classmethod
    def from_omegas(cls, omega1, omega2, omega3, omega4=None) -> "WavesParams":
        return cls(omega=plan_from_omegas(omega1, omega2, omega3, omega4))

    @classmethod
    def from_wavelengths(cls, lambda1_m, lambda2_m, lambda3_m, lambda4_m=None) -> "WavesParams":
        return cls(omega=plan_from_wavelengths(lambda1_m, lambda2_m, lambda3_m, lambda4_m))


@dataclass(frozen=True, slots=True)
class FiberParams:
    length_m: float
    gamma_W_m: float
    alpha_1_m: float = 0.0                               # POWER attenuation; the field decays with alpha/2
    dispersion: Optional[DispersionParams] = None
    beta_legacy_1_m: Optional[np.ndarray] = None         # legacy beta(w_j); fallback dbeta = b3+b4-b1-b2

    def __post_init__(self) -> None:
        object.__setattr__(self, "length_m", _scalar(self.length_m, "length_m", minimum=0.0, strict=True))
        object.__setattr__(self, "gamma_W_m", _scalar(self.gamma_W_m, "gamma_W_m"))
        object.__setattr__(self, "alpha_1_m", _scalar(self.alpha_1_m, "alpha_1_m", minimum=0.0))
        if self.dispersion is not None and not isinstance(self.dispersion, DispersionParams):
            raise TypeError("dispersion must be DispersionParams or None")
        if self.beta_legacy_1_m is not None:
            object.__setattr__(self, "beta_legacy_1_m", _vec4(self.beta_legacy_1_m, "beta_legacy_1_m", positive=False))


@dataclass(frozen=True, slots=True)
class SimulationGrid:
    dz_m: float
    z0_m: float = 0.0

    def __post_init__(self) -> None:
        object.__setattr__(self, "dz_m", _scalar(self.dz_m, "dz_m", minimum=0.0, strict=True))
        object.__setattr__(self, "z0_m", _scalar(self.z0_m, "z0_m"))


@dataclass(frozen=True, slots=True)
class PhaseMatchingParams:
    config: PhaseMatchingConfig

    def __post_init__(self) -> None:
        if not isinstance(self.config, PhaseMatchingConfig):
            raise TypeError("config must be a PhaseMatchingConfig")


@dataclass(slots=True)
class CacheParams:
    """Mutable: filled once per run with the dbeta that exp(+-i*dbeta*z) uses."""
    delta_beta_1_m: Optional[float] = None
    symmetric: Optional[SymmetricPlan] = None

    def set_phase_mismatch(self, delta_beta_1_m: float, symmetric: Optional[SymmetricPlan] = None) -> None:
        self.delta_beta_1_m = _scalar(delta_beta_1_m, "delta_beta_1_m")
        self.symmetric = symmetric


@dataclass(frozen=True, slots=True)
class ModelParams:
    waves: WavesParams
    fiber: FiberParams
    grid: SimulationGrid
    phase_matching: PhaseMatchingParams
    cache: CacheParams

    def __post_init__(self) -> None:
        if not isinstance(self.cache, CacheParams):
            raise TypeError("cache must be a CacheParams (mutable cache object)")


def make_default_phase_matching_params(*, method: PhaseMatchingMethod = PhaseMatchingMethod.SYMMETRIC_EVEN
                                       ) -> PhaseMatchingParams:
    return PhaseMatchingParams(PhaseMatchingConfig(method=method, max_order=4, even_orders=(2, 4), atol=0.0,
                                                   rtol=1e-12))


def make_model_params(*, waves: WavesParams, fiber: FiberParams, grid: SimulationGrid,
                      phase_matching: Optional[PhaseMatchingParams] = None) -> ModelParams:
    pm = phase_matching if phase_matching is not None else make_default_phase_matching_params()
    return ModelParams(waves, fiber, grid, pm, CacheParams(None, waves.symmetric))
